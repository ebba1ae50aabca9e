// vd3d_heal.hip -- heal_missing_pixels (core/render_3d.py:431-459): the gradient-based hole fill north_star names.
// The reference defines the function and never calls it (its loop goes straight from pixel_shift_cuda to the DOF), so here
// it is an optional, default-off stage with its own entry point (vd3d_heal_missing_pixels), bit-exact against the oracle and
// the reference goldens (tests/golden/heal.npz).
//
// One launch, one 64x16 output tile per workgroup, everything between the loads and the store stays in LDS:
//   gray    mean over channels ((r+g)+b)/3 on the tile + halo (1 blur + 2 pool + 1 for the left / top difference)
//   flag    |grad gray| > 0.05  (forward differences, zero in column 0 / row 0)
//   mask    clamp(5x5 box count / 25) -> max with the caller's edge mask                  (tile + 1 halo)
//   healed  (1 - hs*m)*warped + (hs*m)*original                                          (tile + 1 halo, 3 channels)
//   out     (1 - 0.3*m)*healed + (0.3*m)*box3x3(healed)/9, clamp           (row-major running sum: ATen's avg_pool2d order)
// Float32, one rounding per reference operator (-ffp-contract=off), so the result equals the oracle bit for bit.
#include "vd3d_dev.h"
#include "vd3d_kernels.h"

#define HL_TW 64
#define HL_TH 16
#define HL_GW (HL_TW + 7)   // gray  : x0-4 .. x0+TW+2
#define HL_GH (HL_TH + 7)
#define HL_FW (HL_TW + 6)   // flag  : x0-3 .. x0+TW+2
#define HL_FH (HL_TH + 6)
#define HL_MW (HL_TW + 2)   // mask / healed : x0-1 .. x0+TW
#define HL_MH (HL_TH + 2)

__global__ __launch_bounds__(256) void k_heal(const float* __restrict__ warped, const float* __restrict__ orig,
                                              const float* __restrict__ edge, int H, int W, float hs, float* __restrict__ out) {
  __shared__ float gray[HL_GH][HL_GW];
  __shared__ float flag[HL_FH][HL_FW];
  __shared__ float mask[HL_MH][HL_MW];
  __shared__ float healed[3][HL_MH][HL_MW];
  const int x0 = blockIdx.x * HL_TW, y0 = blockIdx.y * HL_TH, tid = threadIdx.x;
  const size_t n = (size_t)H * W;
  for (int t = tid; t < HL_GH * HL_GW; t += 256) {
    const int ty = t / HL_GW, tx = t - ty * HL_GW;
    const int y = y0 - 4 + ty, x = x0 - 4 + tx;
    float g = 0.f;
    if (y >= 0 && y < H && x >= 0 && x < W) {
      const size_t i = (size_t)y * W + x;
      g = ((warped[i] + warped[n + i]) + warped[2 * n + i]) / 3.f;
    }
    gray[ty][tx] = g;
  }
  __syncthreads();
  for (int t = tid; t < HL_FH * HL_FW; t += 256) {
    const int ty = t / HL_FW, tx = t - ty * HL_FW;
    const int y = y0 - 3 + ty, x = x0 - 3 + tx;
    float f = 0.f;                                   // outside the image: avg_pool2d's zero padding
    if (y >= 0 && y < H && x >= 0 && x < W) {
      const float c = gray[ty + 1][tx + 1];
      const float gx = x > 0 ? c - gray[ty + 1][tx] : 0.f;
      const float gy = y > 0 ? c - gray[ty][tx + 1] : 0.f;
      f = vd_sqrt_torch(gx * gx + gy * gy, c_vd_rs14) > (float)0.05 ? 1.f : 0.f;
    }
    flag[ty][tx] = f;
  }
  __syncthreads();
  for (int t = tid; t < HL_MH * HL_MW; t += 256) {
    const int ty = t / HL_MW, tx = t - ty * HL_MW;
    const int y = y0 - 1 + ty, x = x0 - 1 + tx;
    float m = 0.f, h0 = 0.f, h1 = 0.f, h2 = 0.f;
    if (y >= 0 && y < H && x >= 0 && x < W) {
      float sum = 0.f;
#pragma unroll
      for (int dy = 0; dy < 5; ++dy)
#pragma unroll
        for (int dx = 0; dx < 5; ++dx) sum += flag[ty + dy][tx + dx];
      m = vd_clamp(sum / 25.f, 0.f, 1.f);
      const size_t i = (size_t)y * W + x;
      if (edge) { const float e = edge[i]; m = m > e ? m : e; }
      const float a = hs * m, b = 1.0f - a;
      h0 = b * warped[i] + a * orig[i];
      h1 = b * warped[n + i] + a * orig[n + i];
      h2 = b * warped[2 * n + i] + a * orig[2 * n + i];
    }
    mask[ty][tx] = m;
    healed[0][ty][tx] = h0; healed[1][ty][tx] = h1; healed[2][ty][tx] = h2;
  }
  __syncthreads();
  for (int t = tid; t < HL_TH * HL_TW; t += 256) {
    const int ty = t / HL_TW, tx = t - ty * HL_TW;
    const int y = y0 + ty, x = x0 + tx;
    if (y >= H || x >= W) continue;
    const float m = mask[ty + 1][tx + 1];
    const float a = (float)0.3 * m, b = 1.0f - a;
    const size_t i = (size_t)y * W + x;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float sum = 0.f;   // positions outside the image hold 0 and are skipped by ATen: adding +0 leaves the running sum unchanged
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const int yy = y - 1 + dy, xx = x - 1 + dx;
          if (yy >= 0 && yy < H && xx >= 0 && xx < W) sum += healed[c][ty + dy][tx + dx];
        }
      const float v = b * healed[c][ty + 1][tx + 1] + a * (sum / 9.f);
      out[c * n + i] = vd_clamp(v, 0.f, 1.f);
    }
  }
}

void vd_launch_heal(hipStream_t s, const float* warped, const float* orig, const float* edge_or_null, int H, int W, float hs,
                    float* out) {
  dim3 g((W + HL_TW - 1) / HL_TW, (H + HL_TH - 1) / HL_TH);
  hipLaunchKernelGGL(k_heal, g, dim3(256), 0, s, warped, orig, edge_or_null, H, W, hs, out);
}
