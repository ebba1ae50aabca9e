// vd3d_depthprep.hip -- depth-net input preparation (boundary B3): what the HF DPT image processor does between
// a decoded frame and the network (core/render_depth.py:1106-1119 -> transformers DPTImageProcessor): keep-aspect resize to
// a multiple of 14 around 518 with an ANTIALIASED bicubic filter (PIL / torch `antialias=True`, cubic a = -0.5, support
// 2*scale), rescale 1/255, ImageNet normalise -- fused here with the BGR->RGB swap, the bf16 cast and the NHWC
// (channels_last) layout the MIOpen / hipBLASLt kernels of the network want.  One launch replaces ~8 ATen kernels
// (flip, float cast, upsample_gen2d_aa, div, sub, div, bf16 cast, channels_last copy) and their HBM round trips.
//
// Arithmetic follows ATen's upsample_gen2d_aa (float32): weights w_j = cubic((j + xmin - center + 0.5)/scale) normalised
// by their sum, horizontal sums per input row first, then the vertical sum.  Floating-point kernel: tested against the
// torch float32 path on the GPU (tests/test_hip_depthprep.py, bf16-ulp tolerance), no CPU oracle.
#include "vd3d_dev.h"
#include "vd3d_kernels.h"

#define DP_TX 32
#define DP_TY 8
#define DP_KMAX 24   // taps per axis: scale up to ~5.5 (3840 -> 924 needs 18)

struct vd_prep_args {
  int B, H, W, th, tw;
  float scale_h, scale_w;
  float mean[3], stdv[3];
  int in_rows_max, in_cols_max;   // capacity of the input tile (host-computed bound)
};

VD_DEV float dp_cubic(float x) {   // a = -0.5 (PIL-compatible antialias filter)
  const float a = -0.5f;
  x = fabsf(x);
  if (x < 1.f) return ((a + 2.f) * x - (a + 3.f)) * x * x + 1.f;
  if (x < 2.f) return (((x - 5.f) * x + 8.f) * x - 4.f) * a;
  return 0.f;
}
// _compute_weights_aa for output index i: first input index + normalised weights; returns the tap count
VD_DEV int dp_weights(int i, int in, float scale, float* w, int* first) {
  const float support = scale >= 1.f ? 2.f * scale : 2.f;
  const float invscale = scale >= 1.f ? 1.f / scale : 1.f;
  const float center = scale * ((float)i + 0.5f);
  int xmin = (int)(center - support + 0.5f); xmin = xmin < 0 ? 0 : xmin;
  int xend = (int)(center + support + 0.5f); xend = xend > in ? in : xend;
  int n = xend - xmin; n = n > DP_KMAX ? DP_KMAX : n;
  float total = 0.f;
  for (int j = 0; j < n; ++j) { const float v = dp_cubic(((float)(j + xmin) - center + 0.5f) * invscale); w[j] = v; total += v; }
  for (int j = 0; j < n; ++j) w[j] = total != 0.f ? w[j] / total : w[j];
  *first = xmin;
  return n;
}
VD_DEV uint16_t dp_bf16(float f) {   // round-to-nearest-even, finite inputs
  const uint32_t b = __float_as_uint(f);
  return (uint16_t)((b + 0x7fffu + ((b >> 16) & 1u)) >> 16);
}

VD_DEV void dp_store(uint16_t* o, float f) { *o = dp_bf16(f); }
VD_DEV void dp_store(float* o, float f) { *o = f; }

template <typename OT>   // uint16_t = bfloat16 bits, float = float32 (the reference's precision)
__global__ __launch_bounds__(256) void k_depth_prep(const uint8_t* __restrict__ frames, vd_prep_args a, OT* __restrict__ out) {
  extern __shared__ unsigned char smem[];
  float* wx = reinterpret_cast<float*>(smem);                  // [DP_TX][DP_KMAX]
  float* wy = wx + DP_TX * DP_KMAX;                            // [DP_TY][DP_KMAX]
  int* x0s = reinterpret_cast<int*>(wy + DP_TY * DP_KMAX);     // [DP_TX] first input column, [DP_TX] tap count
  int* y0s = x0s + 2 * DP_TX;                                  // [DP_TY] first input row,    [DP_TY] tap count
  float* hb = reinterpret_cast<float*>(y0s + 2 * DP_TY);       // [in_rows_max][DP_TX][3]
  uint8_t* tile = reinterpret_cast<uint8_t*>(hb + (size_t)a.in_rows_max * DP_TX * 3);   // [in_rows_max][in_cols_max*3]
  const int tid = threadIdx.x;
  const int ox0 = blockIdx.x * DP_TX, oy0 = blockIdx.y * DP_TY, b = blockIdx.z;
  if (tid < DP_TX) {
    int f = 0, n = 0;
    if (ox0 + tid < a.tw) n = dp_weights(ox0 + tid, a.W, a.scale_w, wx + tid * DP_KMAX, &f);
    x0s[tid] = f; x0s[DP_TX + tid] = n;
  } else if (tid >= 64 && tid < 64 + DP_TY) {
    const int t = tid - 64;
    int f = 0, n = 0;
    if (oy0 + t < a.th) n = dp_weights(oy0 + t, a.H, a.scale_h, wy + t * DP_KMAX, &f);
    y0s[t] = f; y0s[DP_TY + t] = n;
  }
  __syncthreads();
  const int nx = min(DP_TX, a.tw - ox0), ny = min(DP_TY, a.th - oy0);
  const int c_lo = x0s[0], c_hi = x0s[nx - 1] + x0s[DP_TX + nx - 1];   // [c_lo, c_hi) input columns
  const int r_lo = y0s[0], r_hi = y0s[ny - 1] + y0s[DP_TY + ny - 1];
  const int ncol = min(c_hi - c_lo, a.in_cols_max), nrow = min(r_hi - r_lo, a.in_rows_max);
  const uint8_t* src = frames + (size_t)b * a.H * a.W * 3;
  const int rowbytes = ncol * 3;
  for (int t = tid; t < nrow * rowbytes; t += 256) {
    const int r = t / rowbytes, cb = t - r * rowbytes;
    tile[(size_t)r * a.in_cols_max * 3 + cb] = src[((size_t)(r_lo + r) * a.W + c_lo) * 3 + cb];
  }
  __syncthreads();
  // horizontal pass: (input row, output column) tasks, 3 channels each (BGR bytes -> RGB planes)
  for (int t = tid; t < nrow * DP_TX; t += 256) {
    const int r = t / DP_TX, ox = t - r * DP_TX;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    if (ox < nx) {
      const int n = x0s[DP_TX + ox];
      const uint8_t* p = tile + (size_t)r * a.in_cols_max * 3 + (size_t)(x0s[ox] - c_lo) * 3;
      const float* w = wx + ox * DP_KMAX;
      for (int j = 0; j < n; ++j) {
        const float wj = w[j];
        s0 += (float)p[3 * j + 2] * wj; s1 += (float)p[3 * j + 1] * wj; s2 += (float)p[3 * j] * wj;
      }
    }
    float* h = hb + ((size_t)r * DP_TX + ox) * 3;
    h[0] = s0; h[1] = s1; h[2] = s2;
  }
  __syncthreads();
  const int ox = tid & (DP_TX - 1), oy = tid / DP_TX;
  if (ox < nx && oy < ny) {
    const int n = y0s[DP_TY + oy], rb = y0s[oy] - r_lo;
    const float* w = wy + oy * DP_KMAX;
    float s[3] = {0.f, 0.f, 0.f};
    for (int j = 0; j < n; ++j) {
      const float* h = hb + ((size_t)(rb + j) * DP_TX + ox) * 3;
      const float wj = w[j];
      s[0] += h[0] * wj; s[1] += h[1] * wj; s[2] += h[2] * wj;
    }
    OT* o = out + (((size_t)b * a.th + (oy0 + oy)) * a.tw + (ox0 + ox)) * 3;   // NHWC
#pragma unroll
    for (int c = 0; c < 3; ++c) dp_store(o + c, ((s[c] / 255.0f) - a.mean[c]) / a.stdv[c]);
  }
}

// returns false when the filter footprint exceeds the kernel's tap budget (caller keeps the ATen path)
bool vd_launch_depth_prep(hipStream_t s, const uint8_t* frames, int B, int H, int W, int th, int tw, const float mean[3],
                          const float stdv[3], int dtype, void* out_nhwc) {
  if (dtype != VD3D_DT_BF16 && dtype != VD3D_DT_F32) return false;
  vd_prep_args a;
  a.B = B; a.H = H; a.W = W; a.th = th; a.tw = tw;
  a.scale_h = (float)H / (float)th; a.scale_w = (float)W / (float)tw;
  const float sup_w = a.scale_w >= 1.f ? 2.f * a.scale_w : 2.f, sup_h = a.scale_h >= 1.f ? 2.f * a.scale_h : 2.f;
  if ((int)(2.f * sup_w) + 2 > DP_KMAX || (int)(2.f * sup_h) + 2 > DP_KMAX) return false;
  for (int c = 0; c < 3; ++c) { a.mean[c] = mean[c]; a.stdv[c] = stdv[c]; }
  a.in_cols_max = (int)(a.scale_w * DP_TX + 2.f * sup_w) + 4;
  a.in_rows_max = (int)(a.scale_h * DP_TY + 2.f * sup_h) + 4;
  size_t lds = sizeof(float) * (DP_TX + DP_TY) * DP_KMAX + sizeof(int) * 2 * (DP_TX + DP_TY) +
               sizeof(float) * (size_t)a.in_rows_max * DP_TX * 3 + (size_t)a.in_rows_max * a.in_cols_max * 3;
  lds = (lds + 15) & ~(size_t)15;
  if (lds > 150 * 1024) return false;
  static bool attr[64] = {false};   // per device: the attribute belongs to the device's copy of the code object
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev >= 0 && dev < 64 && !attr[dev]) {
    (void)hipFuncSetAttribute((const void*)k_depth_prep<uint16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)k_depth_prep<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr[dev] = true;
  }
  dim3 g((tw + DP_TX - 1) / DP_TX, (th + DP_TY - 1) / DP_TY, B);
  if (dtype == VD3D_DT_F32) hipLaunchKernelGGL(k_depth_prep<float>, g, dim3(256), lds, s, frames, a, reinterpret_cast<float*>(out_nhwc));
  else hipLaunchKernelGGL(k_depth_prep<uint16_t>, g, dim3(256), lds, s, frames, a, reinterpret_cast<uint16_t*>(out_nhwc));
  return true;
}
