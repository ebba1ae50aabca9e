// vd3d_upscale.hip -- the byte-side glue of the up-scale stage (core/merged_pipeline.py:219-284) and of the depth
// hand-off with an explicit inference size (a24, core/render_depth.py:1914-1917):
//
//   k_resize_cubic_u8   cv2.resize(u8, dsize, interpolation=INTER_CUBIC): OpenCV's fixed-point bicubic (A = -0.75, 11-bit
//                       coefficients, one rounding at the end).  PARITY UNPINNED: cv2 is not in the build image, so the kernel
//                       follows the published algorithm (imgproc resize: coordinate map, interpolateCubic, short coefficients,
//                       (sum + 2^21) >> 22) and is checked against the oracle's two-pass restatement only.
//   k_esr_pre           preprocess_esr (:219-223): BGR u8 crop -> RGB float / 255, planar or channels-last, f32 / bf16 / f16
//   k_esr_post          postprocess_esr (:225-229): clip(0,1) * 255 -> truncate -> BGR u8
//   k_add_weighted_u8   blend_images' cv2.addWeighted (:231-236): round-half-even of a*alpha + b*beta in float32
//
// All four are HBM-bound byte kernels: one thread per output pixel (resize: 16 taps out of L2) or per 4 bytes.
#include "vd3d_dev.h"
#include "vd3d_kernels.h"
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>

// ---- OpenCV INTER_CUBIC, 8-bit ------------------------------------------------------------------------------------------
struct rc_axis { int o[4]; int c[4]; };

// coordinate map + coefficients of ONE output index along an axis of source length n (replicate border by index clamping)
VD_DEV rc_axis rc_axis_make(int d, double scale, int n) {
  rc_axis r;
  float f = (float)(((double)d + 0.5) * scale - 0.5);
  const int s = (int)floorf(f);
  f -= (float)s;
  const float A = -0.75f;
  float w[4];
  w[0] = ((A * (f + 1.f) - 5.f * A) * (f + 1.f) + 8.f * A) * (f + 1.f) - 4.f * A;
  w[1] = ((A + 2.f) * f - (A + 3.f)) * f * f + 1.f;
  w[2] = ((A + 2.f) * (1.f - f) - (A + 3.f)) * (1.f - f) * (1.f - f) + 1.f;
  w[3] = 1.f - w[0] - w[1] - w[2];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int i = s - 1 + k;
    r.o[k] = i < 0 ? 0 : (i > n - 1 ? n - 1 : i);
    float v = rintf(w[k] * 2048.f);                       // saturate_cast<short>(cvRound(.))
    r.c[k] = (int)fminf(fmaxf(v, -32768.f), 32767.f);
  }
  return r;
}

template <int CN>
__global__ __launch_bounds__(256) void k_resize_cubic_u8(const uint8_t* __restrict__ src, int sh, int sw, uint8_t* __restrict__ dst,
                                                         int dh, int dw, double scale_x, double scale_y) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= dw || y >= dh) return;
  const rc_axis ax = rc_axis_make(x, scale_x, sw), ay = rc_axis_make(y, scale_y, sh);
  int acc[CN];
#pragma unroll
  for (int c = 0; c < CN; ++c) acc[c] = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint8_t* row = src + (size_t)ay.o[k] * sw * CN;
    int h[CN];
#pragma unroll
    for (int c = 0; c < CN; ++c) h[c] = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint8_t* p = row + (size_t)ax.o[j] * CN;
#pragma unroll
      for (int c = 0; c < CN; ++c) h[c] += (int)p[c] * ax.c[j];
    }
#pragma unroll
    for (int c = 0; c < CN; ++c) acc[c] += h[c] * ay.c[k];
  }
  uint8_t* o = dst + ((size_t)y * dw + x) * CN;
#pragma unroll
  for (int c = 0; c < CN; ++c) {
    const int v = (acc[c] + (1 << 21)) >> 22;
    o[c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
  }
}

bool vd_launch_resize_cubic_u8(hipStream_t s, const uint8_t* src, int sh, int sw, int cn, uint8_t* dst, int dh, int dw) {
  if (cn != 1 && cn != 3) return false;
  if (sh == dh && sw == dw) {        // cv::resize copies when the sizes agree
    (void)hipMemcpyAsync(dst, src, (size_t)sh * sw * cn, hipMemcpyDeviceToDevice, s);
    return true;
  }
  // cv::resize: inv_scale = dsize / ssize in double, scale = 1. / inv_scale
  const double scale_x = 1.0 / ((double)dw / (double)sw), scale_y = 1.0 / ((double)dh / (double)sh);
  dim3 g((dw + 63) / 64, (dh + 3) / 4);
  if (cn == 1) hipLaunchKernelGGL(k_resize_cubic_u8<1>, g, dim3(256), 0, s, src, sh, sw, dst, dh, dw, scale_x, scale_y);
  else hipLaunchKernelGGL(k_resize_cubic_u8<3>, g, dim3(256), 0, s, src, sh, sw, dst, dh, dw, scale_x, scale_y);
  return true;
}

// ---- cv2.resize(..., interpolation=cv2.INTER_AREA) on 3-channel uint8, the three paths OpenCV takes (the same arithmetic as the fit of
// k_sharp_mux, which is pinned by the finishing-stage goldens): 2x2 -> (sum + 2) >> 2; other integer ratios -> int sum * float(1/area);
// fractional ratios -> ResizeArea_<uchar, float> (per-row float sums); any up-scaling dimension -> the linear machinery in area mode.
__global__ __launch_bounds__(256) void k_resize_area_u8(const uint8_t* __restrict__ src, int sh, int sw, uint8_t* __restrict__ dst, int dh, int dw,
                                                        int mode, int fx, int fy, double sx, double sy) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= dw || y >= dh) return;
  uint8_t* o = dst + ((size_t)y * dw + x) * 3;
  if (mode == 2) {
    int xi, xa0, xa1, yi, yb0, yb1;
    vd_area_lin_coef(sw, dw, x, &xi, &xa0, &xa1);
    vd_area_lin_coef(sh, dh, y, &yi, &yb0, &yb1);
    const int x1 = xi + 1 < sw ? xi + 1 : sw - 1, y1 = yi + 1 < sh ? yi + 1 : sh - 1;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int r0 = (int)src[((size_t)yi * sw + xi) * 3 + c] * xa0 + (int)src[((size_t)yi * sw + x1) * 3 + c] * xa1;
      const int r1 = (int)src[((size_t)y1 * sw + xi) * 3 + c] * xa0 + (int)src[((size_t)y1 * sw + x1) * 3 + c] * xa1;
      const int q = (((yb0 * (r0 >> 4)) >> 16) + ((yb1 * (r1 >> 4)) >> 16) + 2) >> 2;
      o[c] = (uint8_t)(q < 0 ? 0 : (q > 255 ? 255 : q));
    }
  } else if (mode == 1) {
    float ax[VD_AREA_MAXT], ay[VD_AREA_MAXT];
    int x0s, y0s;
    const int nx = vd_area_taps(sw, sx, x, &x0s, ax), ny = vd_area_taps(sh, sy, y, &y0s, ay);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float acc = 0.f;
      for (int j = 0; j < ny; ++j) {
        float h = 0.f;
        for (int k = 0; k < nx; ++k) h = h + (float)src[((size_t)(y0s + j) * sw + x0s + k) * 3 + c] * ax[k];
        acc = acc + h * ay[j];
      }
      o[c] = vd_sat_rne_u8(acc);
    }
  } else {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      int sum = 0;
      for (int j = 0; j < fy; ++j)
        for (int i = 0; i < fx; ++i) sum += src[((size_t)(y * fy + j) * sw + x * fx + i) * 3 + c];
      o[c] = (fx == 2 && fy == 2) ? (uint8_t)((sum + 2) >> 2) : vd_sat_rne_u8((float)sum * (1.f / (float)(fx * fy)));
    }
  }
}

bool vd_launch_resize_area_u8(hipStream_t s, const uint8_t* src, int sh, int sw, uint8_t* dst, int dh, int dw) {
  if (sh == dh && sw == dw) { (void)hipMemcpyAsync(dst, src, (size_t)sh * sw * 3, hipMemcpyDeviceToDevice, s); return true; }
  const double sx = 1.0 / ((double)dw / sw), sy = 1.0 / ((double)dh / sh);
  int mode = (sw % dw || sh % dh) ? 1 : 0;
  if (dw > sw || dh > sh) mode = 2;
  if (mode == 1 && (sx > VD_AREA_MAXT - 2 || sy > VD_AREA_MAXT - 2)) return false;
  hipLaunchKernelGGL(k_resize_area_u8, dim3((dw + 63) / 64, (dh + 3) / 4), dim3(256), 0, s, src, sh, sw, dst, dh, dw, mode,
                     mode == 0 ? sw / dw : 1, mode == 0 ? sh / dh : 1, sx, sy);
  return true;
}

// ---- cv2.resize(src, (dw, dh)) with the default interpolation INTER_LINEAR on 3-channel uint8 (format_3d_output's VR branch,
// core/render_3d.py:846-849): OpenCV's fixed-point path -- coefficients (1 - f, f) * 2048 rounded to nearest-even, horizontal pass in int, vertical
// combine ((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2 (the vector form real builds run; same machinery as the INTER_AREA
// up-scale above, other coefficients).  Horizontal: f is zeroed where the tap would leave the row; vertical: row indices are clipped.  UNPINNED
// (no cv2 in the build image; the published algorithm).
VD_DEV void vd_lin_coef_x(int ssize, int dsize, int d, int* idx, int* a0, int* a1) {
  const double scale = 1.0 / ((double)dsize / ssize);
  float fx = (float)((d + 0.5) * scale - 0.5);
  int sx = (int)floorf(fx);
  fx -= (float)sx;
  if (sx < 0) { fx = 0.f; sx = 0; }
  if (sx >= ssize - 1) { fx = 0.f; sx = ssize - 1; }
  *idx = sx;
  *a0 = (int)rintf((1.f - fx) * 2048.f);
  *a1 = (int)rintf(fx * 2048.f);
}
VD_DEV void vd_lin_coef_y(int ssize, int dsize, int d, int* i0, int* i1, int* b0, int* b1) {
  const double scale = 1.0 / ((double)dsize / ssize);
  float fy = (float)((d + 0.5) * scale - 0.5);
  const int sy = (int)floorf(fy);
  fy -= (float)sy;
  *i0 = sy < 0 ? 0 : (sy > ssize - 1 ? ssize - 1 : sy);
  *i1 = sy + 1 < 0 ? 0 : (sy + 1 > ssize - 1 ? ssize - 1 : sy + 1);
  *b0 = (int)rintf((1.f - fy) * 2048.f);
  *b1 = (int)rintf(fy * 2048.f);
}
__global__ __launch_bounds__(256) void k_resize_linear_u8(const uint8_t* __restrict__ src, int sh, int sw, uint8_t* __restrict__ dst, int dh, int dw) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= dw || y >= dh) return;
  int xi, xa0, xa1, y0, y1, yb0, yb1;
  vd_lin_coef_x(sw, dw, x, &xi, &xa0, &xa1);
  vd_lin_coef_y(sh, dh, y, &y0, &y1, &yb0, &yb1);
  const int x1 = xi + 1 < sw ? xi + 1 : sw - 1;
  uint8_t* o = dst + ((size_t)y * dw + x) * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int r0 = (int)src[((size_t)y0 * sw + xi) * 3 + c] * xa0 + (int)src[((size_t)y0 * sw + x1) * 3 + c] * xa1;
    const int r1 = (int)src[((size_t)y1 * sw + xi) * 3 + c] * xa0 + (int)src[((size_t)y1 * sw + x1) * 3 + c] * xa1;
    const int q = (((yb0 * (r0 >> 4)) >> 16) + ((yb1 * (r1 >> 4)) >> 16) + 2) >> 2;
    o[c] = (uint8_t)(q < 0 ? 0 : (q > 255 ? 255 : q));
  }
}
void vd_launch_resize_linear_u8(hipStream_t s, const uint8_t* src, int sh, int sw, uint8_t* dst, int dh, int dw) {
  if (sh == dh && sw == dw) { (void)hipMemcpyAsync(dst, src, (size_t)sh * sw * 3, hipMemcpyDeviceToDevice, s); return; }   // cv2.resize to the same size copies
  hipLaunchKernelGGL(k_resize_linear_u8, dim3((dw + 63) / 64, (dh + 3) / 4), dim3(256), 0, s, src, sh, sw, dst, dh, dw);
}

// ---- preprocess_esr / postprocess_esr ------------------------------------------------------------------------------------
template <typename T> VD_DEV T esr_cast(float v);
template <> VD_DEV float esr_cast<float>(float v) { return v; }
template <> VD_DEV __hip_bfloat16 esr_cast<__hip_bfloat16>(float v) { return __float2bfloat16(v); }
template <> VD_DEV __half esr_cast<__half>(float v) { return __float2half(v); }

// src: BGR u8 rows of `pitch` bytes; out: RGB, planar [3][h][w] (hwc = 0) or channels-last [h][w][3] (hwc = 1)
template <typename T>
__global__ __launch_bounds__(256) void k_esr_pre(const uint8_t* __restrict__ src, long long pitch, int h, int w, int hwc, T* __restrict__ out) {
  const long long n = (long long)h * w;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const int y = (int)(i / w), x = (int)(i - (long long)y * w);
    const uint8_t* p = src + (long long)y * pitch + 3ll * x;
    const float r = (float)p[2] / 255.f, g = (float)p[1] / 255.f, b = (float)p[0] / 255.f;
    if (hwc) { out[3 * i] = esr_cast<T>(r); out[3 * i + 1] = esr_cast<T>(g); out[3 * i + 2] = esr_cast<T>(b); }
    else { out[i] = esr_cast<T>(r); out[n + i] = esr_cast<T>(g); out[2 * n + i] = esr_cast<T>(b); }
  }
}

VD_DEV uint8_t esr_u8(float v) {   // np.clip(v, 0, 1) * 255.0 -> astype(uint8): truncation; NaN -> 0
  v = v < 0.f ? 0.f : (v > 1.f ? 1.f : v);
  v = v * 255.f;
  return v == v ? (uint8_t)(int)v : (uint8_t)0;
}

// pred: RGB float32 planar / channels-last of size h x w; the (cy, cx, ch, cw) window of it goes to dst rows of `pitch` bytes
__global__ __launch_bounds__(256) void k_esr_post(const float* __restrict__ pred, int h, int w, int hwc, int cy, int cx, int ch, int cw,
                                                  uint8_t* __restrict__ dst, long long pitch) {
  const long long n = (long long)ch * cw, np = (long long)h * w;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const int y = (int)(i / cw), x = (int)(i - (long long)y * cw);
    const long long s = (long long)(cy + y) * w + (cx + x);
    float r, g, b;
    if (hwc) { r = pred[3 * s]; g = pred[3 * s + 1]; b = pred[3 * s + 2]; }
    else { r = pred[s]; g = pred[np + s]; b = pred[2 * np + s]; }
    uint8_t* o = dst + (long long)y * pitch + 3ll * x;
    o[0] = esr_u8(b); o[1] = esr_u8(g); o[2] = esr_u8(r);
  }
}

// run_rife's glue (core/merged_pipeline.py:195-218): concatenate_images + preprocess_rife = the two frames / 255 stacked to six
// channels in the frames' own BGR order (no swap there), planar [6][h][w] or channels-last [h][w][6]; the output side is
// clip(0,1) * 255 truncated with the channel order untouched.
template <typename T>
__global__ __launch_bounds__(256) void k_rife_pre(const uint8_t* __restrict__ f1, const uint8_t* __restrict__ f2, int h, int w, int hwc,
                                                  T* __restrict__ out) {
  const long long n = (long long)h * w;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      const float v = (float)(c < 3 ? f1[3 * i + c] : f2[3 * i + c - 3]) / 255.f;
      if (hwc) out[6 * i + c] = esr_cast<T>(v); else out[c * n + i] = esr_cast<T>(v);
    }
  }
}
__global__ __launch_bounds__(256) void k_rife_post(const float* __restrict__ pred, int h, int w, int hwc, uint8_t* __restrict__ dst) {
  const long long n = (long long)h * w;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
#pragma unroll
    for (int c = 0; c < 3; ++c) dst[3 * i + c] = esr_u8(hwc ? pred[3 * i + c] : pred[c * n + i]);
}
bool vd_launch_rife_pre(hipStream_t s, int dtype, const uint8_t* f1, const uint8_t* f2, int h, int w, int hwc, void* out) {
  const long long n = (long long)h * w;
  const int g = (int)((n + 255) / 256 > 65535 * 16 ? 65535 * 16 : (n + 255) / 256);
  if (dtype == VD3D_DT_F32) hipLaunchKernelGGL(k_rife_pre<float>, dim3(g), dim3(256), 0, s, f1, f2, h, w, hwc, (float*)out);
  else if (dtype == VD3D_DT_BF16) hipLaunchKernelGGL(k_rife_pre<__hip_bfloat16>, dim3(g), dim3(256), 0, s, f1, f2, h, w, hwc, (__hip_bfloat16*)out);
  else if (dtype == VD3D_DT_F16) hipLaunchKernelGGL(k_rife_pre<__half>, dim3(g), dim3(256), 0, s, f1, f2, h, w, hwc, (__half*)out);
  else return false;
  return true;
}
void vd_launch_rife_post(hipStream_t s, const float* pred, int h, int w, int hwc, uint8_t* dst) {
  const long long n = (long long)h * w;
  const int g = (int)((n + 255) / 256 > 65535 * 16 ? 65535 * 16 : (n + 255) / 256);
  hipLaunchKernelGGL(k_rife_post, dim3(g), dim3(256), 0, s, pred, h, w, hwc, dst);
}

bool vd_launch_esr_pre(hipStream_t s, int dtype, const uint8_t* src, long long pitch, int h, int w, int hwc, void* out) {
  const long long n = (long long)h * w;
  const int g = (int)((n + 255) / 256 > 65535 * 16 ? 65535 * 16 : (n + 255) / 256);
  if (dtype == VD3D_DT_F32) hipLaunchKernelGGL(k_esr_pre<float>, dim3(g), dim3(256), 0, s, src, pitch, h, w, hwc, (float*)out);
  else if (dtype == VD3D_DT_BF16) hipLaunchKernelGGL(k_esr_pre<__hip_bfloat16>, dim3(g), dim3(256), 0, s, src, pitch, h, w, hwc, (__hip_bfloat16*)out);
  else if (dtype == VD3D_DT_F16) hipLaunchKernelGGL(k_esr_pre<__half>, dim3(g), dim3(256), 0, s, src, pitch, h, w, hwc, (__half*)out);
  else return false;
  return true;
}

void vd_launch_esr_post(hipStream_t s, const float* pred, int h, int w, int hwc, int cy, int cx, int ch, int cw, uint8_t* dst, long long pitch) {
  const long long n = (long long)ch * cw;
  const int g = (int)((n + 255) / 256 > 65535 * 16 ? 65535 * 16 : (n + 255) / 256);
  hipLaunchKernelGGL(k_esr_post, dim3(g), dim3(256), 0, s, pred, h, w, hwc, cy, cx, ch, cw, dst, pitch);
}

// ---- cv2.addWeighted(a, alpha, b, beta, 0) on uint8 ------------------------------------------------------------------------
// float32: fma(a, alpha, b * beta + gamma), rounded half-to-even, saturated (the v_fma form of OpenCV's SIMD arithm kernel).
__global__ __launch_bounds__(256) void k_add_weighted_u8(const uint8_t* __restrict__ a, float alpha, const uint8_t* __restrict__ b, float beta,
                                                         float gamma, long long n, uint8_t* __restrict__ out) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float t = __builtin_fmaf((float)a[i], alpha, __builtin_fmaf((float)b[i], beta, gamma));
    const float r = rintf(t);
    out[i] = (uint8_t)(r < 0.f ? 0 : (r > 255.f ? 255 : (int)r));
  }
}

void vd_launch_add_weighted_u8(hipStream_t s, const uint8_t* a, float alpha, const uint8_t* b, float beta, float gamma, long long n, uint8_t* out) {
  const int g = (int)((n + 255) / 256 > 65535 * 16 ? 65535 * 16 : (n + 255) / 256);
  hipLaunchKernelGGL(k_add_weighted_u8, dim3(g), dim3(256), 0, s, a, alpha, b, beta, gamma, n, out);
}
