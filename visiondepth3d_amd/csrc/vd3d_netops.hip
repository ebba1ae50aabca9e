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

// float32 variant (the reference runs its HF depth models in float32, core/render_depth.py:758-759): x + y is ATen's exact float32 add,
// LayerNorm statistics two-pass in float32.  One wave per row, float2 per lane per step (cols = NJ * 128).
template <int NJ>
__global__ __launch_bounds__(256) void k_add_layernorm_f32(const float2* __restrict__ x, const float2* __restrict__ y,
                                                           const float2* __restrict__ gamma, const float2* __restrict__ beta,
                                                           float eps, long long rows, float2* __restrict__ out_sum,
                                                           float2* __restrict__ out_norm) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const size_t base = (size_t)row * (NJ * 64);   // in float2
  float v[2 * NJ];
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    float2 a = x[base + lane + 64 * j];
    if (y) {
      const float2 b = y[base + lane + 64 * j];
      a.x += b.x; a.y += b.y;
      out_sum[base + lane + 64 * j] = a;
    }
    v[2 * j] = a.x; v[2 * j + 1] = a.y;
    sum += a.x + a.y;
  }
  const float n = (float)(NJ * 128);
  const float mean = wave_sum_f(sum) / n;
  float sq = 0.f;
#pragma unroll
  for (int j = 0; j < 2 * NJ; ++j) { const float d = v[j] - mean; sq += d * d; }
  const float rstd = 1.0f / sqrtf(wave_sum_f(sq) / n + eps);
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const float2 g = gamma[lane + 64 * j], b = beta[lane + 64 * j];
    out_norm[base + lane + 64 * j] = make_float2((v[2 * j] - mean) * rstd * g.x + b.x, (v[2 * j + 1] - mean) * rstd * g.y + b.y);
  }
}

bool vd_launch_add_layernorm(hipStream_t s, int dtype, const void* x, const void* y, const void* gamma, const void* beta, float eps,
                             long long rows, int cols, void* out_sum, void* out_norm) {
  const dim3 g((unsigned)((rows + 3) / 4)), b(256);
  if (dtype == VD3D_DT_F32) {
    const float2 *xx = (const float2*)x, *yy = (const float2*)y, *gg = (const float2*)gamma, *bb = (const float2*)beta;
    float2 *os = (float2*)out_sum, *on = (float2*)out_norm;
    switch (cols) {
      case 384: hipLaunchKernelGGL(k_add_layernorm_f32<3>, g, b, 0, s, xx, yy, gg, bb, eps, rows, os, on); return true;
      case 768: hipLaunchKernelGGL(k_add_layernorm_f32<6>, g, b, 0, s, xx, yy, gg, bb, eps, rows, os, on); return true;
      case 1024: hipLaunchKernelGGL(k_add_layernorm_f32<8>, g, b, 0, s, xx, yy, gg, bb, eps, rows, os, on); return true;
      default: return false;
    }
  }
  if (dtype != VD3D_DT_BF16) return false;
  const uint32_t *xx = (const uint32_t*)x, *yy = (const uint32_t*)y, *gg = (const uint32_t*)gamma, *bb = (const uint32_t*)beta;
  uint32_t *os = (uint32_t*)out_sum, *on = (uint32_t*)out_norm;
  switch (cols) {
    case 384: hipLaunchKernelGGL(k_add_layernorm<3>, g, b, 0, s, xx, yy, gg, bb, eps, rows, os, on); return true;
    case 768: hipLaunchKernelGGL(k_add_layernorm<6>, g, b, 0, s, xx, yy, gg, bb, eps, rows, os, on); return true;
    case 1024: hipLaunchKernelGGL(k_add_layernorm<8>, g, b, 0, s, xx, yy, gg, bb, eps, rows, os, on); return true;
    default: return false;
  }
}

// k_upsample_bilinear_nhwc: F.interpolate(mode="bilinear", align_corners=True) on a bf16 channels_last tensor (the five
// up-samplings of the DPT neck / head).  One thread = 8 channels (16 B) of one output pixel; float32 blend in ATen's
// association (l0y*(l0x*p00 + l1x*p01) + l1y*(l0x*p10 + l1x*p11)), rounded once to bf16.  ATen's own NHWC kernel runs at
// ~1/7 of the HBM rate on these shapes (measured 1.78 ms per 16-frame batch for ~1 GB of traffic).
__global__ __launch_bounds__(256) void k_upsample_bilinear_nhwc(const uint4* __restrict__ in, uint4* __restrict__ out, int ih, int iw,
                                                                int oh, int ow, int c8, float sh, float sw, long long total) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int c = (int)(idx % c8);
  long long r = idx / c8;
  const int x = (int)(r % ow); r /= ow;
  const int y = (int)(r % oh);
  const long long b = r / oh;
  const float fy = sh * (float)y, fx = sw * (float)x;
  const int y0 = (int)fy, x0 = (int)fx;
  const int y1 = y0 + (y0 < ih - 1 ? 1 : 0), x1 = x0 + (x0 < iw - 1 ? 1 : 0);
  const float ly1 = fy - (float)y0, ly0 = 1.f - ly1, lx1 = fx - (float)x0, lx0 = 1.f - lx1;
  const uint4* base = in + (size_t)b * ih * iw * c8 + c;
  const uint4 p00 = base[((size_t)y0 * iw + x0) * c8], p01 = base[((size_t)y0 * iw + x1) * c8];
  const uint4 p10 = base[((size_t)y1 * iw + x0) * c8], p11 = base[((size_t)y1 * iw + x1) * c8];
  const uint32_t a00[4] = {p00.x, p00.y, p00.z, p00.w}, a01[4] = {p01.x, p01.y, p01.z, p01.w};
  const uint32_t a10[4] = {p10.x, p10.y, p10.z, p10.w}, a11[4] = {p11.x, p11.y, p11.z, p11.w};
  uint32_t o[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float lo = ly0 * (lx0 * bf2f(a00[j] & 0xffffu) + lx1 * bf2f(a01[j] & 0xffffu)) +
                     ly1 * (lx0 * bf2f(a10[j] & 0xffffu) + lx1 * bf2f(a11[j] & 0xffffu));
    const float hi = ly0 * (lx0 * bf2f(a00[j] >> 16) + lx1 * bf2f(a01[j] >> 16)) +
                     ly1 * (lx0 * bf2f(a10[j] >> 16) + lx1 * bf2f(a11[j] >> 16));
    o[j] = f2bf(lo) | (f2bf(hi) << 16);
  }
  out[idx] = make_uint4(o[0], o[1], o[2], o[3]);
}
// float32 variant: one thread = 4 channels (16 B), ATen's association
__global__ __launch_bounds__(256) void k_upsample_bilinear_nhwc_f32(const float4* __restrict__ in, float4* __restrict__ out, int ih, int iw,
                                                                    int oh, int ow, int c4, float sh, float sw, long long total) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int c = (int)(idx % c4);
  long long r = idx / c4;
  const int x = (int)(r % ow); r /= ow;
  const int y = (int)(r % oh);
  const long long b = r / oh;
  const float fy = sh * (float)y, fx = sw * (float)x;
  const int y0 = (int)fy, x0 = (int)fx;
  const int y1 = y0 + (y0 < ih - 1 ? 1 : 0), x1 = x0 + (x0 < iw - 1 ? 1 : 0);
  const float ly1 = fy - (float)y0, ly0 = 1.f - ly1, lx1 = fx - (float)x0, lx0 = 1.f - lx1;
  const float4* base = in + (size_t)b * ih * iw * c4 + c;
  const float4 p00 = base[((size_t)y0 * iw + x0) * c4], p01 = base[((size_t)y0 * iw + x1) * c4];
  const float4 p10 = base[((size_t)y1 * iw + x0) * c4], p11 = base[((size_t)y1 * iw + x1) * c4];
  float4 o;
  o.x = ly0 * (lx0 * p00.x + lx1 * p01.x) + ly1 * (lx0 * p10.x + lx1 * p11.x);
  o.y = ly0 * (lx0 * p00.y + lx1 * p01.y) + ly1 * (lx0 * p10.y + lx1 * p11.y);
  o.z = ly0 * (lx0 * p00.z + lx1 * p01.z) + ly1 * (lx0 * p10.z + lx1 * p11.z);
  o.w = ly0 * (lx0 * p00.w + lx1 * p01.w) + ly1 * (lx0 * p10.w + lx1 * p11.w);
  out[idx] = o;
}
bool vd_launch_upsample_bilinear_nhwc(hipStream_t s, int dtype, const void* in, void* out, int B, int ih, int iw, int oh, int ow, int C) {
  if (dtype == VD3D_DT_F32) {
    if (C % 4 || oh < 2 || ow < 2) return false;
    const int c4 = C / 4;
    const long long total = (long long)B * oh * ow * c4;
    const float sh = (float)(ih - 1) / (float)(oh - 1), sw = (float)(iw - 1) / (float)(ow - 1);
    hipLaunchKernelGGL(k_upsample_bilinear_nhwc_f32, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const float4*)in, (float4*)out,
                       ih, iw, oh, ow, c4, sh, sw, total);
    return true;
  }
  if (dtype != VD3D_DT_BF16) return false;
  if (C % 8 || oh < 2 || ow < 2) return false;
  const int c8 = C / 8;
  const long long total = (long long)B * oh * ow * c8;
  const float sh = (float)(ih - 1) / (float)(oh - 1), sw = (float)(iw - 1) / (float)(ow - 1);   // area_pixel_compute_scale, align_corners
  hipLaunchKernelGGL(k_upsample_bilinear_nhwc, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const uint4*)in, (uint4*)out, ih, iw,
                     oh, ow, c8, sh, sw, total);
  return true;
}

// ---- DPT neck / head glue in float32 (round 4): what sits between MIOpen's convolutions in DepthAnythingPreActResidualLayer /
// FeatureFusionLayer / DepthEstimationHead (transformers modeling_depth_anything.py).  PyTorch runs each bias, ReLU and residual sum as its
// own pass over the feature map (bias: a broadcast add behind every MIOpen convolution); at 4K the head-resolution maps are 0.6 - 1.3 GB.
//
// k_bias_act: v = y [+ bias[c]] [+ r1] ; [v = r2 + v] ; [v = max(v, 0)] -> out ; [relu_out = max(v, 0)]   (NHWC, one thread = 4 channels)
//   second convolution of a residual unit: bias + the unit's input (+ the fusion layer's running state) in one pass, and the ReLU'd copy the next
//   unit's first convolution reads; first convolution: bias + ReLU in place.
__global__ __launch_bounds__(256) void k_bias_act(const float4* __restrict__ y, const float4* __restrict__ bias, const float4* __restrict__ r1,
                                                  const float4* __restrict__ r2, int relu, int c4, long long total, float4* __restrict__ out,
                                                  float4* __restrict__ relu_out) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  float4 v = y[idx];
  if (bias) { const float4 b = bias[(int)(idx % c4)]; v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
  if (r1) { const float4 r = r1[idx]; v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
  if (r2) { const float4 r = r2[idx]; v.x = r.x + v.x; v.y = r.y + v.y; v.z = r.z + v.z; v.w = r.w + v.w; }
  if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
  out[idx] = v;
  if (relu_out) relu_out[idx] = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
}
bool vd_launch_bias_act_f32(hipStream_t s, const float* y, const float* bias, const float* r1, const float* r2, int relu, long long n_pix, int C,
                            float* out, float* relu_out) {
  if (C % 4 || n_pix < 1) return false;
  const long long total = n_pix * (C / 4);
  hipLaunchKernelGGL(k_bias_act, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const float4*)y, (const float4*)bias, (const float4*)r1,
                     (const float4*)r2, relu, C / 4, total, (float4*)out, (float4*)relu_out);
  return true;
}

// k_upsample_bilinear_nhwc_f32 with the producing convolution's bias added to the interpolated value: the interpolation weights sum to one,
// so up(conv + b) == up(conv) + b; the convolution in front runs without its bias pass.
__global__ __launch_bounds__(256) void k_upsample_bilinear_bias_nhwc_f32(const float4* __restrict__ in, const float4* __restrict__ bias,
                                                                         float4* __restrict__ out, int ih, int iw, int oh, int ow, int c4, float sh,
                                                                         float sw, long long total) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int c = (int)(idx % c4);
  long long r = idx / c4;
  const int x = (int)(r % ow); r /= ow;
  const int y = (int)(r % oh);
  const long long b = r / oh;
  const float fy = sh * (float)y, fx = sw * (float)x;
  const int y0 = (int)fy, x0 = (int)fx;
  const int y1 = y0 + (y0 < ih - 1 ? 1 : 0), x1 = x0 + (x0 < iw - 1 ? 1 : 0);
  const float ly1 = fy - (float)y0, ly0 = 1.f - ly1, lx1 = fx - (float)x0, lx0 = 1.f - lx1;
  const float4* base = in + (size_t)b * ih * iw * c4 + c;
  const float4 p00 = base[((size_t)y0 * iw + x0) * c4], p01 = base[((size_t)y0 * iw + x1) * c4];
  const float4 p10 = base[((size_t)y1 * iw + x0) * c4], p11 = base[((size_t)y1 * iw + x1) * c4];
  const float4 bb = bias[c];
  float4 o;
  o.x = (ly0 * (lx0 * p00.x + lx1 * p01.x) + ly1 * (lx0 * p10.x + lx1 * p11.x)) + bb.x;
  o.y = (ly0 * (lx0 * p00.y + lx1 * p01.y) + ly1 * (lx0 * p10.y + lx1 * p11.y)) + bb.y;
  o.z = (ly0 * (lx0 * p00.z + lx1 * p01.z) + ly1 * (lx0 * p10.z + lx1 * p11.z)) + bb.z;
  o.w = (ly0 * (lx0 * p00.w + lx1 * p01.w) + ly1 * (lx0 * p10.w + lx1 * p11.w)) + bb.w;
  out[idx] = o;
}
bool vd_launch_upsample_bilinear_bias_nhwc_f32(hipStream_t s, const float* in, const float* bias, float* out, int B, int ih, int iw, int oh, int ow, int C) {
  if (C % 4 || oh < 2 || ow < 2 || !bias) return false;
  const int c4 = C / 4;
  const long long total = (long long)B * oh * ow * c4;
  const float sh = (float)(ih - 1) / (float)(oh - 1), sw = (float)(iw - 1) / (float)(ow - 1);
  hipLaunchKernelGGL(k_upsample_bilinear_bias_nhwc_f32, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const float4*)in, (const float4*)bias,
                     (float4*)out, ih, iw, oh, ow, c4, sh, sw, total);
  return true;
}

// k_head_tail: everything behind the head's second convolution -- its bias, ReLU, the 1x1 convolution to ONE channel (a C-term dot product per
// pixel), its bias, ReLU and max_depth -- in one pass: reads the C-channel map once, writes one float per pixel (PyTorch: bias r+w, ReLU r+w,
// MIOpen 1x1 r, bias, ReLU, scale).  LPP = C / 4 adjacent lanes share a pixel (coalesced 16-byte loads), partial dot products meet in a butterfly.
template <int LPP>
__global__ __launch_bounds__(256) void k_head_tail(const float4* __restrict__ y, const float4* __restrict__ b2, const float4* __restrict__ w3, float b3,
                                                   float scale, long long n_pix, float* __restrict__ out) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long pix = idx / LPP;
  const int l = (int)(idx % LPP);
  float acc = 0.f;
  if (pix < n_pix) {
    const float4 v = y[idx], b = b2[l], w = w3[l];
    acc = fmaxf(v.x + b.x, 0.f) * w.x;
    acc = vd_fma(fmaxf(v.y + b.y, 0.f), w.y, acc);
    acc = vd_fma(fmaxf(v.z + b.z, 0.f), w.z, acc);
    acc = vd_fma(fmaxf(v.w + b.w, 0.f), w.w, acc);
  }
#pragma unroll
  for (int off = LPP / 2; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if (l == 0 && pix < n_pix) out[pix] = fmaxf(acc + b3, 0.f) * scale;
}
bool vd_launch_head_tail_f32(hipStream_t s, const float* y, const float* b2, const float* w3, float b3, float scale, long long n_pix, int C, float* out) {
  if (n_pix < 1) return false;
  const long long total = n_pix * (C / 4);
  const dim3 g((unsigned)((total + 255) / 256)), b(256);
  const float4 *yy = (const float4*)y, *bb = (const float4*)b2, *ww = (const float4*)w3;
  switch (C) {
    case 16: hipLaunchKernelGGL(k_head_tail<4>, g, b, 0, s, yy, bb, ww, b3, scale, n_pix, out); return true;
    case 32: hipLaunchKernelGGL(k_head_tail<8>, g, b, 0, s, yy, bb, ww, b3, scale, n_pix, out); return true;
    case 64: hipLaunchKernelGGL(k_head_tail<16>, g, b, 0, s, yy, bb, ww, b3, scale, n_pix, out); return true;
    default: return false;
  }
}
