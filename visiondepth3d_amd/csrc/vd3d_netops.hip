// vd3d_netops.hip -- memory-bound glue kernels of the depth network's transformer blocks (boundary B3; the GEMMs,
// attention and convolutions stay in hipBLASLt / AOTriton / MIOpen as north_star prescribes).
//
// k_add_layernorm:  s = x + y (bf16, rounded like ATen's add), out_norm = LayerNorm(s) * gamma + beta (float32 statistics,
// biased variance, eps inside the sqrt), one wave per row, two rows of traffic instead of ATen's add kernel + layer-norm
// kernel (5 rows).  y == nullptr -> plain LayerNorm.  cols must be a multiple of 128 (2 bf16 per lane per step).
#include "vd3d_dev.h"
#include "vd3d_kernels.h"

VD_DEV float bf2f(uint32_t h) { return __uint_as_float(h << 16); }
VD_DEV uint32_t f2bf(float f) { const uint32_t b = __float_as_uint(f); return (b + 0x7fffu + ((b >> 16) & 1u)) >> 16; }
VD_DEV float wave_sum_f(float v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

template <int NJ>   // cols = NJ * 128
__global__ __launch_bounds__(256) void k_add_layernorm(const uint32_t* __restrict__ x, const uint32_t* __restrict__ y,
                                                       const uint32_t* __restrict__ gamma, const uint32_t* __restrict__ beta,
                                                       float eps, long long rows, uint32_t* __restrict__ out_sum,
                                                       uint32_t* __restrict__ out_norm) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const size_t base = (size_t)row * (NJ * 64);   // in dwords (2 bf16 each)
  float v[2 * NJ];
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const uint32_t a = x[base + lane + 64 * j];
    float lo = bf2f(a & 0xffffu), hi = bf2f(a >> 16);
    if (y) {
      const uint32_t b = y[base + lane + 64 * j];
      const uint32_t slo = f2bf(lo + bf2f(b & 0xffffu)), shi = f2bf(hi + bf2f(b >> 16));
      out_sum[base + lane + 64 * j] = slo | (shi << 16);
      lo = bf2f(slo); hi = bf2f(shi);
    }
    v[2 * j] = lo; v[2 * j + 1] = hi;
    sum += lo + hi;
  }
  const float n = (float)(NJ * 128);
  const float mean = wave_sum_f(sum) / n;
  float sq = 0.f;
#pragma unroll
  for (int j = 0; j < 2 * NJ; ++j) { const float d = v[j] - mean; sq += d * d; }
  const float rstd = 1.0f / sqrtf(wave_sum_f(sq) / n + eps);
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const uint32_t g = gamma[lane + 64 * j], b = beta[lane + 64 * j];
    const float lo = (v[2 * j] - mean) * rstd * bf2f(g & 0xffffu) + bf2f(b & 0xffffu);
    const float hi = (v[2 * j + 1] - mean) * rstd * bf2f(g >> 16) + bf2f(b >> 16);
    out_norm[base + lane + 64 * j] = f2bf(lo) | (f2bf(hi) << 16);
  }
}

bool vd_launch_add_layernorm(hipStream_t s, const void* x, const void* y, const void* gamma, const void* beta, float eps,
                             long long rows, int cols, void* out_sum, void* out_norm) {
  const dim3 g((unsigned)((rows + 3) / 4)), b(256);
  const uint32_t *xx = (const uint32_t*)x, *yy = (const uint32_t*)y, *gg = (const uint32_t*)gamma, *bb = (const uint32_t*)beta;
  uint32_t *os = (uint32_t*)out_sum, *on = (uint32_t*)out_norm;
  switch (cols) {
    case 384: hipLaunchKernelGGL(k_add_layernorm<3>, g, b, 0, s, xx, yy, gg, bb, eps, rows, os, on); return true;
    case 768: hipLaunchKernelGGL(k_add_layernorm<6>, g, b, 0, s, xx, yy, gg, bb, eps, rows, os, on); return true;
    case 1024: hipLaunchKernelGGL(k_add_layernorm<8>, g, b, 0, s, xx, yy, gg, bb, eps, rows, os, on); return true;
    default: return false;
  }
}
