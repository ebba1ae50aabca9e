// vd3d_dev.h -- device-side arithmetic primitives for the gfx950 DIBR kernels.
//
// Arithmetic contract (DESIGN.md "Numerics"): float32, one rounding per reference operator, built
// with -ffp-contract=off; FMA only where written explicitly (the places ATen's own fused kernels
// contract); pow/exp are the correctly-rounded float32 value computed through float64.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vd3d.h"

#define VD_DEV __device__ __forceinline__

// XCD-aware tile order (MI355X: 8 XCDs, each with a private 4 MB L2; workgroup b is observed to run on XCD b % 8 -- a speed
// assumption only, never a correctness one).  A 1-D grid of 8 * per workgroups walks the tiles so that every XCD owns one
// contiguous band of `per` tiles in row-major order: neighbouring tiles' halo re-reads then hit the XCD's own L2 instead of being
// fetched once per XCD through the fabric.  Returns the tile index (>= the tile count for the padding workgroups of the last band).
VD_DEV int vd_xcd_tile(int b, int per, int on) { return on ? (b & 7) * per + (b >> 3) : b; }
// The same idea for kernels whose cost per tile follows the image content (E1: the DOF levels a tile needs): contiguous bands would
// give every XCD one region of the picture and unbalance them (measured: +18 % kernel time), so the tile rows are dealt out in groups
// of G rows, group g to XCD g % 8; inside a group the walk is row-major, so the horizontal halo and the halo between the group's rows
// stay in one L2.  Returns row = -1 ... the caller checks row < rows.  Grid: 8 * ceil(ceil(rows / G) / 8) * G * ntx workgroups.
VD_DEV void vd_xcd_tile_rows(int b, int ntx, int G, int on, int* row, int* col) {
  if (!on) { *row = b / ntx; *col = b - *row * ntx; return; }
  const int x = b & 7, k = b >> 3, pg = G * ntx;
  const int gl = k / pg, within = k - gl * pg, wr = within / ntx;
  *row = (gl * 8 + x) * G + wr;
  *col = within - wr * ntx;
}

VD_DEV float vd_clamp(float x, float lo, float hi) {
  float t = x < lo ? lo : x;
  return t > hi ? hi : t;
}
VD_DEV float vd_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
// clamp for FINITE x and lo < hi as one v_med3_f32 (vd_clamp needs 2 compares + 2 selects to keep NaN/-0 semantics).
// Differs from vd_clamp only in NaN propagation and in the sign of a zero result; used where neither can reach the output
// (pixel arithmetic that ends in a uint8 truncation), never for histogram keys.
VD_DEV float vd_clamp_fin(float x, float lo, float hi) { return __builtin_amdgcn_fmed3f(x, lo, hi); }

// torch.linspace float32, element form: step=(end-start)/(steps-1); i<steps/2 ? fma(step,i,start) : fma(-step,steps-1-i,end)
VD_DEV float vd_linspace(float start, float end, int steps, int i) {
  if (steps == 1) return start;
  float step = (end - start) / (float)(steps - 1);
  if (i < steps / 2) return vd_fma(step, (float)i, start);
  return vd_fma(-step, (float)(steps - 1 - i), end);
}
VD_DEV float vd_lin11(int steps, int i) { return vd_linspace(-1.f, 1.f, steps, i); }
// same value with the (loop-invariant) step hoisted: step = (1 - (-1)) / (float)(steps - 1)
VD_DEV float vd_lin11_step(float step, int steps, int i) {
  if (steps == 1) return -1.f;
  if (i < steps / 2) return vd_fma(step, (float)i, -1.f);
  return vd_fma(-step, (float)(steps - 1 - i), 1.f);
}
// correctly-rounded float32 pow / exp through float64 (device libm is < 1 ULP in float64)
VD_DEV float vd_pow_cr(float x, float e) { return (float)pow((double)x, (double)e); }
VD_DEV float vd_exp_cr(float x) { return (float)exp((double)x); }
// Table-driven float64 pow for x in (0, 1]: log2 by a 128-entry table + degree-7 polynomial, 2^f by a 64-entry table + degree-6
// polynomial, ~30 float64 operations instead of libm's ~200.  Relative error < 2^-45; whenever the float64 result lies within
// 2^-39 of a float32 rounding boundary (3e-5 of the inputs) the function reports "ambiguous" and the caller uses vd_pow_cr, so the
// value is ALWAYS the float32 rounding of the libm result.  tools/verify_fastpow.c restates these exact operations on the CPU and
// checks every float in (0, 1] for g = 0.85 (the only exponent render_sbs_3d passes), 0.3, 0.5, 0.999, 1, 1.5, 2.2: 0 mismatches.
// tab: [0,128) 1/c_i, [128,256) log2 c_i, [256,320) 2^(j/64)  (vd3d_pow_tables.h; the kernels stage it in LDS).
VD_DEV bool vd_pow_fast(float x, double g, const double* __restrict__ tab, float* out) {
  const uint32_t b = __float_as_uint(x);
  if (b < 0x00800000u || b > 0x3f800000u) return false;          // zero, subnormal, > 1: exact path
  const int e = (int)(b >> 23) - 127;
  const int i = (int)((b >> 16) & 0x7fu);
  const double m = (double)__uint_as_float((b & 0x007fffffu) | 0x3f800000u);
  const double r = __builtin_fma(m, tab[i], -1.0);               // |r| <= 2^-8
  const double C1 = 1.4426950408889634074, C2 = -0.72134752044448170368, C3 = 0.48089834696298780245, C4 = -0.36067376022224085184,
               C5 = 0.28853900817779268147, C6 = -0.24044917348149390123, C7 = 0.20609929155556620106;
  double p = __builtin_fma(r, C7, C6); p = __builtin_fma(r, p, C5); p = __builtin_fma(r, p, C4); p = __builtin_fma(r, p, C3);
  p = __builtin_fma(r, p, C2); p = __builtin_fma(r, p, C1);
  const double L = ((double)e + tab[128 + i]) + r * p;
  const double y = g * L;
  if (!(y > -120.0)) return false;
  const double k = __builtin_rint(y * 64.0);
  const double f = __builtin_fma(k, -1.0 / 64.0, y);             // |f| <= 2^-7
  const long long ki = (long long)k;
  const int j = (int)(ki & 63);
  const int n = (int)((ki - j) / 64);
  const double t = f * 0.69314718055994530942;
  double q = __builtin_fma(t, 1.0 / 6.0, 1.0); q = __builtin_fma(t * (1.0 / 5.0), q, 1.0); q = __builtin_fma(t * (1.0 / 4.0), q, 1.0);
  q = __builtin_fma(t * (1.0 / 3.0), q, 1.0); q = __builtin_fma(t * 0.5, q, 1.0); q = __builtin_fma(t, q, 1.0);
  const double res = ldexp(tab[256 + j] * q, n);
  const uint32_t low = (uint32_t)((unsigned long long)__double_as_longlong(res) & 0x1fffffffull);   // mantissa bits below float32 precision
  const uint32_t half = 0x10000000u;
  const uint32_t dist = low > half ? low - half : half - low;
  if (dist < (1u << 13)) return false;                           // within 2^-39 of a rounding boundary
  *out = (float)res;
  return true;
}
VD_DEV float vd_pow_cr_fast(float x, float e, const double* __restrict__ tab) {   // == vd_pow_cr(x, e) for every x
  float v;
  if (vd_pow_fast(x, (double)e, tab, &v)) return v;
  return vd_pow_cr(x, e);
}

// x^1.5 == x*sqrt(x): float64 sqrt is correctly rounded, product error < 1 ULP(float64) => same float32 as pow
VD_DEV float vd_pow15_cr(float x) {
  double d = (double)x;
  return (float)(d * sqrt(d));
}

// packed-f32 vector types: hipcc lowers arithmetic on these to v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 (2 lanes-worth of
// IEEE float32 work per VALU issue slot on gfx950); each element is rounded exactly like the scalar operator.
typedef float vd_f2 __attribute__((ext_vector_type(2)));
typedef float vd_f4 __attribute__((ext_vector_type(4)));
VD_DEV float vd_vfma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
VD_DEV vd_f2 vd_vfma(vd_f2 a, vd_f2 b, vd_f2 c) { return __builtin_elementwise_fma(a, b, c); }
VD_DEV vd_f4 vd_vfma(vd_f4 a, vd_f4 b, vd_f4 c) { return __builtin_elementwise_fma(a, b, c); }

// (float)v / 255.0f for an integer v in [0,255] without the division: one Newton residual step makes the product
// correctly rounded; verified exhaustively for all 256 inputs (tests/test_host_logic.py restates the check).
VD_DEV float vd_u8_unit(float x) {
  const float rc = 1.0f / 255.0f;
  const float q0 = x * rc;
  return vd_fma(vd_fma(-q0, 255.0f, x), rc, q0);
}

// DOF Gaussian tap sum: the kernel is exactly symmetric, taps are paired outermost-first and
// accumulated with fused multiply-adds:  acc = w0*(v0+v[K-1]); acc = fma(w_t, v_t+v[K-1-t], acc) ...; acc = fma(w_r, v_r, acc)
template <int K, typename V>
VD_DEV V vd_gauss_sym(const float* __restrict__ kw, const V* v) {
  constexpr int r = K / 2;
  V acc = kw[0] * (v[0] + v[K - 1]);
#pragma unroll
  for (int t = 1; t < r; ++t) acc = vd_vfma((V)(kw[t]), v[t] + v[K - 1 - t], acc);
  return vd_vfma((V)(kw[r]), v[r], acc);
}
VD_DEV float vd_gauss_sym_rt(const float* __restrict__ kw, int k, const float* v, int stride) {  // run-time tap count
  const int r = k / 2;
  float acc = kw[0] * (v[0] + v[(k - 1) * stride]);
  for (int t = 1; t < r; ++t) acc = vd_fma(kw[t], v[t * stride] + v[(k - 1 - t) * stride], acc);
  return vd_fma(kw[r], v[r * stride], acc);
}

// F.interpolate(bilinear, align_corners=False) tap for output index o
struct vd_tap { int i0, i1; float w0, w1; };
VD_DEV vd_tap vd_interp_tap(int in, int out, int o) {
  vd_tap t;
  if (in == out) { t.i0 = o; t.i1 = o; t.w0 = 1.f; t.w1 = 0.f; return t; }
  const float scale = (float)in / (float)out;
  float src = vd_fma(scale, (float)o + 0.5f, -0.5f);   // area_pixel_compute_source_index: ONE fused multiply-add in ATen's builds
  if (src < 0.f) src = 0.f;
  int i0 = (int)floorf(src);
  if (i0 > in - 1) i0 = in - 1;
  float l1 = vd_clamp(src - (float)i0, 0.f, 1.f);
  t.i0 = i0;
  t.i1 = i0 + (i0 < in - 1 ? 1 : 0);
  t.w1 = l1;
  t.w0 = 1.f - l1;
  return t;
}
VD_DEV vd_tap vd_interp_tap_s(int in, int out, float scale, int o) {  // vd_interp_tap with scale = (float)in/(float)out hoisted
  vd_tap t;
  if (in == out) { t.i0 = o; t.i1 = o; t.w0 = 1.f; t.w1 = 0.f; return t; }
  float src = vd_fma(scale, (float)o + 0.5f, -0.5f);   // area_pixel_compute_source_index: ONE fused multiply-add in ATen's builds
  if (src < 0.f) src = 0.f;
  int i0 = (int)floorf(src);
  if (i0 > in - 1) i0 = in - 1;
  float l1 = vd_clamp(src - (float)i0, 0.f, 1.f);
  t.i0 = i0; t.i1 = i0 + (i0 < in - 1 ? 1 : 0); t.w1 = l1; t.w0 = 1.f - l1;
  return t;
}

// vd_interp_tap(in, 2*in, o) in closed form: scale = 0.5 exactly, src = 0.5*(o + 0.5) - 0.5 = 0.5*o - 0.25 (exact in float32), clamped
// at 0 for o = 0; even o = 2m: i0 = m-1, l1 = 0.75; odd o = 2m+1: i0 = m, l1 = 0.25.  Same values as the general routine.
VD_DEV vd_tap vd_tap21(int in, int o) {
  vd_tap t;
  if (o == 0) { t.i0 = 0; t.w1 = 0.f; }
  else { t.i0 = (o - 1) >> 1; t.w1 = (o & 1) ? 0.25f : 0.75f; }
  t.i1 = t.i0 + (t.i0 < in - 1 ? 1 : 0);
  t.w0 = 1.f - t.w1;
  return t;
}

// ATen Interpolate<>::eval association: fma(t0,w0,t1*w1), rows then columns
VD_DEV float vd_bilerp(float p00, float p01, float p10, float p11, float wx0, float wx1, float wy0, float wy1) {
  float a = vd_fma(p00, wx0, wx1 * p01);
  float b = vd_fma(p10, wx0, wx1 * p11);
  return vd_fma(a, wy0, wy1 * b);
}

// grid_sample(bilinear, border, align_corners=True) parameters for a normalised coordinate pair
struct vd_gs { int xw, yn; float nw, ne, sw, se; bool e_ok, s_ok; };
VD_DEV vd_gs vd_gs_params(float gx, float gy, int W, int H) {
  vd_gs p;
  float ix = (gx + 1.f) * ((float)(W - 1) / 2.f);
  float iy = (gy + 1.f) * ((float)(H - 1) / 2.f);
  ix = fminf((float)(W - 1), fmaxf(ix, 0.f));
  iy = fminf((float)(H - 1), fmaxf(iy, 0.f));
  float xw = floorf(ix), yn = floorf(iy);
  float w = ix - xw, e = 1.f - w, n = iy - yn, s = 1.f - n;
  p.nw = s * e; p.ne = s * w; p.sw = n * e; p.se = n * w;
  p.xw = (int)xw; p.yn = (int)yn;
  p.e_ok = (p.xw + 1) < W;
  p.s_ok = (p.yn + 1) < H;
  return p;
}
VD_DEV float vd_gs_combine(const vd_gs& p, float vnw, float vne, float vsw, float vse) {
  return vd_fma(vse, p.se, vd_fma(vsw, p.sw, vd_fma(vne, p.ne, vnw * p.nw)));
}

// order-independent fixed-point accumulation (oracle: fx40)
#define VD_FX 1099511627776.0
VD_DEV long long vd_fx40(double v) { return __double2ll_rn(v * VD_FX); }

VD_DEV int vd_reflect(int i, int n) {
  if (n == 1) return 0;
  while (i < 0 || i >= n) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
  }
  return i;
}
VD_DEV uint8_t vd_sat_rne_u8(float v) {
  return (uint8_t)__builtin_amdgcn_fmed3f(rintf(v), 0.f, 255.f);   // finite v (sums of uint8 * weights)
}

// 64-bit wave reductions (wave = 64 lanes on gfx950)
VD_DEV long long vd_wave_sum_ll(long long v) {
  for (int off = 32; off > 0; off >>= 1) {
    int lo = __shfl_down((int)(v & 0xffffffffll), off, 64);
    int hi = __shfl_down((int)(v >> 32), off, 64);
    v += ((long long)hi << 32) | (unsigned int)lo;
  }
  return v;
}

// histogram add with wave-level aggregation of equal keys (smooth depth planes put most of a wave in
// one bin; a plain atomic would serialise 64-way on that address)
template <typename HistPtr>
VD_DEV void vd_hist_add_agg(HistPtr hist, unsigned key, bool valid) {
  unsigned long long mask = __ballot(valid);
  const int lane = threadIdx.x & 63;
  // at most two leader rounds (a constant or two-valued plane collapses to <= 2 atomics per wave); whatever is left has
  // many distinct keys, where one plain atomic per lane is cheaper than one ballot/shuffle round per key
#pragma unroll
  for (int round = 0; round < 2; ++round) {
    if (!mask) return;
    const int leader = __ffsll((long long)mask) - 1;
    const unsigned lk = __shfl((int)key, leader, 64);
    const unsigned long long same = __ballot(valid && key == lk) & mask;
    if (lane == leader) atomicAdd(&hist[lk], (unsigned)__popcll(same));
    mask &= ~same;
  }
  if ((mask >> lane) & 1ull) atomicAdd(&hist[key], 1u);
}

// LDS histogram add: a plain returnless ds_add_u32.  Same-address lanes serialise at ~1 lane/clk inside the LDS, so
// even a fully uniform wave costs ~64 cycles -- cheaper than one round of the ballot/shuffle aggregation above, which
// is therefore reserved for GLOBAL atomics (pass B), where every conflicting lane would be a separate L2 operation.
VD_DEV void vd_lds_hist_add(uint32_t* hist, unsigned key, bool valid) {
  if (valid) atomicAdd(&hist[key], 1u);
}

// ---- cv2.resize INTER_AREA coefficient helpers (shared by the fit of k_sharp_mux and the standalone k_resize_area_u8)
// hal::resize linear coefficients in "area mode" (INTER_AREA with an up-scaling dimension), one destination index
VD_DEV void vd_area_lin_coef(int ssize, int dsize, int d, int* idx, int* a0, int* a1) {
  const double inv = (double)dsize / ssize, scale = 1.0 / inv;
  int sx = (int)floor(d * scale);
  float fx = (float)((d + 1) - (sx + 1) * inv);
  fx = fx <= 0 ? 0.f : fx - floorf(fx);
  if (sx < 0) { fx = 0.f; sx = 0; }
  if (sx >= ssize - 1) { fx = 0.f; sx = ssize - 1; }
  *idx = sx;
  *a0 = (int)rintf((1.f - fx) * 2048.f);
  *a1 = (int)rintf(fx * 2048.f);
}
// computeResizeAreaTab (OpenCV resize.cpp) for ONE destination index: consecutive source indices s0..s0+n-1 with weights a[].
#define VD_AREA_MAXT 12
VD_DEV int vd_area_taps(int ssize, double scale, int d, int* s0, float* a) {
  const double fsx1 = d * scale, fsx2 = fsx1 + scale;
  const double cell = scale < (double)ssize - fsx1 ? scale : (double)ssize - fsx1;
  int sx1 = (int)ceil(fsx1), sx2 = (int)floor(fsx2);
  sx2 = sx2 < ssize - 1 ? sx2 : ssize - 1;
  sx1 = sx1 < sx2 ? sx1 : sx2;
  int n = 0;
  *s0 = sx1;
  if (sx1 - fsx1 > 1e-3) { *s0 = sx1 - 1; a[n++] = (float)((sx1 - fsx1) / cell); }
  for (int sx = sx1; sx < sx2 && n < VD_AREA_MAXT; ++sx) a[n++] = (float)(1.0 / cell);
  if (fsx2 - sx2 > 1e-3 && n < VD_AREA_MAXT) {
    double t = fsx2 - sx2; t = t < 1.0 ? t : 1.0; t = t < cell ? t : cell;
    a[n++] = (float)(t / cell);
  }
  return n;
}
