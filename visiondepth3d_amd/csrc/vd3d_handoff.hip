// vd3d_handoff.hip -- depth hand-off: depth-net output -> the uint8 depth plane the DIBR stage reads.
//
// Reference: transformers' depth-estimation post-process (bicubic F.interpolate to the frame size,
// align_corners=False) followed by convert_depth_to_grayscale (core/render_depth.py:585-611: per-frame min-max,
// (norm*255).astype(uint8) truncation, optional 255-u8 at :1914-1916), then an XVID depth video on disk.
// Here: two small passes per batch, no full-resolution float plane and no disk hop:
//   pass 1  bicubic value at every output pixel -> wave/block min-max -> one atomicMin/Max per workgroup on
//           order-preserving uint keys (exact, order-independent)
//   pass 2  bicubic again (the 1.9 MB prediction stays in L2) -> normalise -> truncate -> packed uint8 stores
// Arithmetic = oracle/vd3d_oracle.c:vo_depth_handoff (float32, fixed association, no contraction).
#include "vd3d_dev.h"
#include "vd3d_kernels.h"

struct vd_handoff_args { int B, ph, pw, H, W, invert, same; float sh, sw; };

VD_DEV float cubic1(float x, float A) { return ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f; }
VD_DEV float cubic2(float x, float A) { return ((A * x - 5.f * A) * x + 8.f * A) * x - 4.f * A; }
VD_DEV void cubic_coeffs(float t, float c[4]) {
  const float A = -0.75f;
  c[0] = cubic2(t + 1.f, A); c[1] = cubic1(t, A); c[2] = cubic1(1.f - t, A); c[3] = cubic2((1.f - t) + 1.f, A);
}
VD_DEV float bicubic_at(const float* __restrict__ p, const vd_handoff_args& a, const float cy[4], int iy, int x) {
  const float rx = vd_fma(a.sw, (float)x + 0.5f, -0.5f);   // fused source index, like the bilinear taps (vd3d_dev.h)
  const float fx = floorf(rx);
  const int ix = (int)fx;
  float cx[4];
  cubic_coeffs(rx - fx, cx);
  int xs[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { int xx = ix - 1 + j; xs[j] = xx < 0 ? 0 : (xx > a.pw - 1 ? a.pw - 1 : xx); }
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int yy = iy - 1 + i; yy = yy < 0 ? 0 : (yy > a.ph - 1 ? a.ph - 1 : yy);
    const float* row = p + (size_t)yy * a.pw;
    float r = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) r += row[xs[j]] * cx[j];
    acc += r * cy[i];
  }
  return acc;
}
VD_DEV uint32_t f2key(float v) { uint32_t b = __float_as_uint(v); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
VD_DEV float key2f(uint32_t k) { uint32_t b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k; return __uint_as_float(b); }

__global__ void k_handoff_init(uint32_t* mm, int B) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B) { mm[3 * i] = 0xffffffffu; mm[3 * i + 1] = 0u; mm[3 * i + 2] = 0u; }
}

#define HO_ROWS 32  // rows per workgroup (4 waves x 8 rows each): ~8k pixels per atomic pair in pass 1
template <bool WRITE>
__global__ __launch_bounds__(256) void k_handoff(const float* __restrict__ pred, vd_handoff_args a, uint32_t* __restrict__ mm,
                                                 uint8_t* __restrict__ out) {
  const int b = blockIdx.z;
  const float* p = pred + (size_t)b * a.ph * a.pw;
  const int xq = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;  // 4 consecutive pixels per thread
  float bmn = INFINITY, bmx = -INFINITY;
  int bbad = 0;
  for (int ry_ = 0; ry_ < HO_ROWS / 4; ++ry_) {
  const int y = blockIdx.y * HO_ROWS + ry_ * 4 + (threadIdx.x >> 6);
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  const bool row_ok = y < a.H;
  if (row_ok && xq < a.W) {
    if (a.same) {
#pragma unroll
      for (int q = 0; q < 4; ++q) if (xq + q < a.W) v[q] = p[(size_t)y * a.W + xq + q];
    } else {
      const float ry = vd_fma(a.sh, (float)y + 0.5f, -0.5f);
      const float fy = floorf(ry);
      float cy[4];
      cubic_coeffs(ry - fy, cy);
#pragma unroll
      for (int q = 0; q < 4; ++q) if (xq + q < a.W) v[q] = bicubic_at(p, a, cy, (int)fy, xq + q);
    }
  }
  if (!WRITE) {
    float mn = INFINITY, mx = -INFINITY;
    int bad = 0;
    if (row_ok)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (xq + q < a.W) { bad |= (v[q] != v[q]); mn = v[q] < mn ? v[q] : mn; mx = v[q] > mx ? v[q] : mx; }
    bmn = fminf(bmn, mn); bmx = fmaxf(bmx, mx); bbad |= bad;
  } else {
    if (!row_ok || xq >= a.W) continue;
    const float mn = key2f(mm[3 * b]), mx = key2f(mm[3 * b + 1]);
    const bool flat = mm[3 * b + 2] != 0u || (mx - mn) < (float)1e-6;
    const float den = (mx - mn) + (float)1e-6;
    uint32_t pack = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      uint8_t u = 0;
      if (!flat) {
        float n = ((v[q] - mn) / den) * 255.f;
        n = n < 0.f ? 0.f : (n > 255.f ? 255.f : n);
        u = (uint8_t)n;
      }
      if (a.invert) u = (uint8_t)(255 - u);
      pack |= (uint32_t)u << (8 * q);
    }
    uint8_t* o = out + (size_t)b * a.H * a.W + (size_t)y * a.W + xq;
    if (xq + 3 < a.W && (((size_t)b * a.H * a.W + (size_t)y * a.W + xq) & 3) == 0) *reinterpret_cast<uint32_t*>(o) = pack;
    else for (int q = 0; q < 4 && xq + q < a.W; ++q) o[q] = (uint8_t)(pack >> (8 * q));
  }
  }  // rows
  if (!WRITE) {
    __shared__ float smn[4], smx[4];
    __shared__ int sbad[4];
    for (int off = 32; off > 0; off >>= 1) {
      bmn = fminf(bmn, __shfl_down(bmn, off, 64));
      bmx = fmaxf(bmx, __shfl_down(bmx, off, 64));
      bbad |= __shfl_down(bbad, off, 64);
    }
    if ((threadIdx.x & 63) == 0) { smn[threadIdx.x >> 6] = bmn; smx[threadIdx.x >> 6] = bmx; sbad[threadIdx.x >> 6] = bbad; }
    __syncthreads();
    if (threadIdx.x == 0) {
      const float mn = fminf(fminf(smn[0], smn[1]), fminf(smn[2], smn[3]));
      const float mx = fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
      if (mn <= mx) { atomicMin(&mm[3 * b], f2key(mn)); atomicMax(&mm[3 * b + 1], f2key(mx)); }
      if (sbad[0] | sbad[1] | sbad[2] | sbad[3]) atomicOr(&mm[3 * b + 2], 1u);
    }
  }
}

void vd_launch_depth_handoff(hipStream_t s, const float* pred, int B, int ph, int pw, int H, int W, int invert, uint32_t* mm,
                             uint8_t* out) {
  vd_handoff_args a;
  a.B = B; a.ph = ph; a.pw = pw; a.H = H; a.W = W; a.invert = invert; a.same = (ph == H && pw == W);
  a.sh = (float)ph / (float)H; a.sw = (float)pw / (float)W;
  hipLaunchKernelGGL(k_handoff_init, dim3((B + 63) / 64), dim3(64), 0, s, mm, B);
  dim3 g((W + 255) / 256, (H + HO_ROWS - 1) / HO_ROWS, B);
  hipLaunchKernelGGL(k_handoff<false>, g, dim3(256), 0, s, pred, a, mm, out);
  hipLaunchKernelGGL(k_handoff<true>, g, dim3(256), 0, s, pred, a, mm, out);
}
