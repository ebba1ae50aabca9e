// vd3d_heatmap.hip -- the colour-mapped previews of generate_preview_image (core/preview_utils.py:42-66):
//   0 "Shift Heatmap"                cv2.normalize(shift, None, 0, 255, NORM_MINMAX).astype(uint8)          -> applyColorMap
//   1 "Shift Heatmap (Abs)"          the same on |shift|
//   2 "Shift Heatmap (Clipped ±5px)" ((clip(shift, -5, 5) + 5) / 10 * 255).astype(uint8)                    -> applyColorMap
//   3 "Feather Mask"                 clip(|shift| * 50, 0, 255).astype(uint8)                               -> applyColorMap
// The index arithmetic is float32 in the reference's operator order; the 256-entry BGR table is the CALLER's (OpenCV's COLORMAP_JET /
// COLORMAP_BONE tables are data of that library, not restated here).  Types 0 / 1 need the plane's min and max first: one reduction
// launch on order-preserving uint keys (exact, order-independent), then the map launch.
// cv2.normalize: scale = 255 * (smax - smin > DBL_EPSILON ? 1 / (smax - smin) : 0) and shift = -smin * scale in double, then
// convertTo(float32) = fma(src, (float)scale, (float)shift) (the v_fma form of OpenCV's SIMD cvtScale); PARITY UNPINNED for that
// rounding (cv2 is not in the build image) -- the other two types are plain numpy arithmetic.
#include "vd3d_dev.h"
#include "vd3d_kernels.h"

VD_DEV uint32_t hm_key(float v) { uint32_t b = __float_as_uint(v); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
VD_DEV float hm_unkey(uint32_t k) { uint32_t b = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k; return __uint_as_float(b); }

__global__ void k_hm_init(uint32_t* mm) { mm[0] = 0xffffffffu; mm[1] = 0u; mm[2] = 0u; }

__global__ __launch_bounds__(256) void k_hm_minmax(const float* __restrict__ s, long long n, int use_abs, uint32_t* __restrict__ mm) {
  float mn = INFINITY, mx = -INFINITY;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    float v = s[i];
    if (use_abs) v = fabsf(v);
    mn = v < mn ? v : mn; mx = v > mx ? v : mx;      // NaNs are skipped (cv::minMaxIdx compares with <, >)
  }
  for (int off = 32; off > 0; off >>= 1) { mn = fminf(mn, __shfl_down(mn, off, 64)); mx = fmaxf(mx, __shfl_down(mx, off, 64)); }
  __shared__ float smn[4], smx[4];
  if ((threadIdx.x & 63) == 0) { smn[threadIdx.x >> 6] = mn; smx[threadIdx.x >> 6] = mx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    mn = fminf(fminf(smn[0], smn[1]), fminf(smn[2], smn[3]));
    mx = fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
    if (mn <= mx) { atomicMin(&mm[0], hm_key(mn)); atomicMax(&mm[1], hm_key(mx)); }
  }
}

VD_DEV uint8_t hm_astype_u8(float v) { return (uint8_t)(int)v; }   // numpy's float32 -> uint8 cast: truncation (low byte out of range)

__global__ __launch_bounds__(256) void k_hm_map(const float* __restrict__ s, long long n, int type, const uint32_t* __restrict__ mm,
                                                const uint8_t* __restrict__ lut, uint8_t* __restrict__ out) {
  __shared__ uint8_t l[768];
  for (int t = threadIdx.x; t < 768; t += 256) l[t] = lut[t];
  __syncthreads();
  float a = 0.f, b = 0.f;
  if (type <= 1) {
    const double smin = (double)hm_unkey(mm[0]), smax = (double)hm_unkey(mm[1]);
    const double scale = 255.0 * (smax - smin > 2.220446049250313e-16 ? 1.0 / (smax - smin) : 0.0);
    a = (float)scale; b = (float)(0.0 - smin * scale);
  }
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    float v = s[i];
    uint8_t idx;
    if (type == 0) idx = hm_astype_u8(__builtin_fmaf(v, a, b));
    else if (type == 1) idx = hm_astype_u8(__builtin_fmaf(fabsf(v), a, b));
    else if (type == 2) {
      float c = v < -5.f ? -5.f : (v > 5.f ? 5.f : v);         // np.clip keeps NaN; the cast below sends it to 0 like x86 does
      idx = hm_astype_u8(((c + 5.f) / 10.f) * 255.f);
    } else {
      float c = fabsf(v) * 50.f;
      c = c < 0.f ? 0.f : (c > 255.f ? 255.f : c);
      idx = hm_astype_u8(c);
    }
    out[3 * i] = l[3 * idx]; out[3 * i + 1] = l[3 * idx + 1]; out[3 * i + 2] = l[3 * idx + 2];
  }
}

bool vd_launch_preview_heatmap(hipStream_t s, int type, const float* shift, int h, int w, const uint8_t* lut_dev, uint32_t* mm, uint8_t* out) {
  if (type < 0 || type > 3) return false;
  const long long n = (long long)h * w;
  const int g = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
  if (type <= 1) {
    hipLaunchKernelGGL(k_hm_init, dim3(1), dim3(1), 0, s, mm);
    hipLaunchKernelGGL(k_hm_minmax, dim3(g), dim3(256), 0, s, shift, n, type == 1, mm);
  }
  hipLaunchKernelGGL(k_hm_map, dim3(g), dim3(256), 0, s, shift, n, type, mm, lut_dev, out);
  return true;
}

// "Overlay Arrows" (core/preview_utils.py:74-82): on a copy of the left eye, every 20th pixel of every 20th row gets
// cv2.arrowedLine((x, y) -> (x + dx, y), green, thickness 1, tipLength 0.3) with dx = int(shift * 10) when |dx| > 1.
// The arrows are horizontal, so OpenCV's rasteriser reduces to closed forms: the shaft is the pixel run between the end points; the two
// tip strokes start at p = pt2 + round(0.3 |dx| (cos, sin)(angle +- pi/4)) with angle = atan2(0, -dx) in {0, pi}, i.e. at
// (x2 -+ r, y -+ r) with r = round(0.3 |dx| / sqrt 2) -- checked for every |dx| <= 4000: both coordinates round to the same r, 7.6e-5
// away from a tie -- and an exact 45-degree Bresenham line is its diagonal.  Lines are clipped to the image (cv::clipLine), which for
// these slopes equals dropping the outside pixels.  All arrows have one colour, so overlapping writes need no order.
// PARITY UNPINNED (cv2 is not in the build image); the closed forms follow drawing.cpp's arrowedLine / LineIterator.
__global__ __launch_bounds__(256) void k_preview_arrows(const float* __restrict__ shift, int h, int w, int gw, int gn, uint8_t* __restrict__ out) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= gn) return;
  const int gy = t / gw, gx = t - gy * gw;
  const int y = 20 * gy, x = 20 * gx;
  const float sv = shift[(size_t)y * w + x] * 10.f;
  if (!(sv == sv) || fabsf(sv) >= 2.0e9f) return;       // int(nan) / int(inf) raise in the reference
  const int dx = (int)sv;                                // Python int(): truncation toward zero
  if (dx > -2 && dx < 2) return;
  const int x2 = x + dx, adx = dx < 0 ? -dx : dx, sg = dx < 0 ? -1 : 1;
  auto put = [&](int py, int px) {
    if (py >= 0 && py < h && px >= 0 && px < w) { uint8_t* o = out + ((size_t)py * w + px) * 3; o[0] = 0; o[1] = 255; o[2] = 0; }
  };
  for (int i = 0; i <= adx; ++i) put(y, x + sg * i);
  const int r = (int)rint((double)adx * 0.3 * 0.70710678118654757);
  for (int i = 0; i <= r; ++i) { put(y - i, x2 - sg * i); put(y + i, x2 - sg * i); }
}

void vd_launch_preview_arrows(hipStream_t s, const uint8_t* left, const float* shift, int h, int w, uint8_t* out) {
  (void)hipMemcpyAsync(out, left, (size_t)h * w * 3, hipMemcpyDeviceToDevice, s);
  const int gw = (w + 19) / 20, gh = (h + 19) / 20, gn = gw * gh;
  hipLaunchKernelGGL(k_preview_arrows, dim3((gn + 255) / 256), dim3(256), 0, s, shift, h, w, gw, gn, out);
}
