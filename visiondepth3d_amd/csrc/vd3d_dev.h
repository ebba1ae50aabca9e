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
// correctly-rounded float32 exp through float64 (device libm is < 1 ULP in float64): torch.exp on the k-tap Gaussian window
VD_DEV float vd_exp_cr(float x) { return (float)exp((double)x); }

// Development aid (compiled out unless -DVD_PHASE_STAMPS: bash tools/build_ab.sh stamps -DVD_PHASE_STAMPS): thread 0 of every 67th workgroup (64 slots)
// records s_memtime (shader cycles) at phase boundaries and s_memrealtime (100 MHz) at stamp 0 and at the last stamp, so that a probe can
// print cycles per phase and the shader clock the kernel actually ran at (tools/probe_phases.py).  One table per translation unit.
#ifdef VD_PHASE_STAMPS
#define VD_STAMP_DECL(name) static __device__ unsigned long long name[64][16]
#define VD_STAMP(name, k, last)                                                                                            \
  do {                                                                                                                     \
    if (threadIdx.x == 0 && blockIdx.x % 67 == 7 && blockIdx.x / 67 < 64) {                                              \
      name[blockIdx.x / 67][k] = __builtin_amdgcn_s_memtime();                                                            \
      if ((k) == 0) name[blockIdx.x / 67][14] = __builtin_amdgcn_s_memrealtime();                                          \
      if (last) name[blockIdx.x / 67][15] = __builtin_amdgcn_s_memrealtime();                                              \
    }                                                                                                                      \
  } while (0)
// residency probe: EVERY workgroup records where (HW_ID, XCC_ID) and when (s_memrealtime at entry / exit of thread 0) it ran
#define VD_OCC_DECL(name, getter)                                                                                          \
  static __device__ unsigned long long name[16384][4];                                                                     \
  extern "C" __attribute__((visibility("default"))) int getter(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(name), sizeof(name)); }
#define VD_OCC_IN(name)                                                                                                    \
  do {                                                                                                                     \
    if (threadIdx.x == 0 && blockIdx.x < 16384) {                                                                          \
      name[blockIdx.x][0] = __builtin_amdgcn_s_getreg((31 << 11) | 4);                                                     \
      name[blockIdx.x][1] = __builtin_amdgcn_s_getreg((31 << 11) | 20);                                                    \
      name[blockIdx.x][2] = __builtin_amdgcn_s_memrealtime();                                                              \
    }                                                                                                                      \
  } while (0)
#define VD_OCC_OUT(name) do { if (threadIdx.x == 0 && blockIdx.x < 16384) name[blockIdx.x][3] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define VD_STAMP_DECL(name)
#define VD_STAMP(name, k, last) do { } while (0)
#define VD_OCC_DECL(name, getter)
#define VD_OCC_IN(name) do { } while (0)
#define VD_OCC_OUT(name) do { } while (0)
#endif

// ---- torch-CPU transcendental numerics, reproduced bit for bit ----------------------------------------------------------------
// The reference runs on ATen's CPU kernels: torch.pow(tensor, float) = SLEEF Sleef_powf_u10, torch.sigmoid = 1 / (1 + Sleef_expf_u10(0 - x)),
// torch.sqrt = MKL VML vsSqrt (one fused correction step on the AVX-512 VRSQRT14 estimate -- NOT the correctly rounded root).  These
// are float32-only algorithms (double-float arithmetic on FMAs), so they run at full VALU rate here; every operation below is one
// IEEE float32 rounding (the library is built with -ffp-contract=off; 1.0f / x and the fmas are exact-rounded on gfx950, float32
// denormals are on).  oracle/vd3d_oracle.c holds the independent C restatement that is pinned against torch itself; the parity tests
// compare the two bit for bit.  Published algorithm: SLEEF 3.6 sleefsimdsp.c (xpowf -> logkf / expkf, xexpf) and df.h.
struct vd_df { float x, y; };
VD_DEV vd_df df_norm(vd_df t) { vd_df s; s.x = t.x + t.y; s.y = t.x - s.x + t.y; return s; }
VD_DEV vd_df df_add2_ff(float x, float y) { vd_df s; s.x = x + y; const float v = s.x - x; s.y = (x - (s.x - v)) + (y - v); return s; }
VD_DEV vd_df df_add_22(vd_df x, vd_df y) { vd_df s; s.x = x.x + y.x; s.y = x.x - s.x + y.x + x.y + y.y; return s; }
VD_DEV vd_df df_add2_2f(vd_df x, float y) { vd_df s; s.x = x.x + y; const float v = s.x - x.x; s.y = (x.x - (s.x - v)) + (y - v); s.y = s.y + x.y; return s; }
VD_DEV vd_df df_add2_22(vd_df x, vd_df y) { vd_df s; s.x = x.x + y.x; const float v = s.x - x.x; s.y = (x.x - (s.x - v)) + (y.x - v); s.y = s.y + (x.y + y.y); return s; }
VD_DEV vd_df df_add_f2(float x, vd_df y) { vd_df s; s.x = x + y.x; s.y = x - s.x + y.x + y.y; return s; }
VD_DEV vd_df df_mul_22(vd_df x, vd_df y) { vd_df r; r.x = x.x * y.x; r.y = vd_fma(x.x, y.y, vd_fma(x.y, y.x, vd_fma(x.x, y.x, -r.x))); return r; }
VD_DEV vd_df df_mul_2f(vd_df x, float y) { vd_df r; r.x = x.x * y; r.y = vd_fma(x.y, y, vd_fma(x.x, y, -r.x)); return r; }
VD_DEV vd_df df_squ(vd_df x) { vd_df r; r.x = x.x * x.x; r.y = vd_fma(x.x + x.x, x.y, vd_fma(x.x, x.x, -r.x)); return r; }
VD_DEV vd_df df_div(vd_df n, vd_df d) {
  const float t = 1.0f / d.x, s = n.x * t;
  const float u = vd_fma(t, n.x, -s);
  const float v = vd_fma(-d.y, t, vd_fma(-d.x, t, 1.0f));
  vd_df r; r.x = s; r.y = vd_fma(s, v, vd_fma(n.y, t, u)); return r;
}
VD_DEV vd_df vd_sleef_logkf(float d) {   // d > 0
  const int e = __builtin_amdgcn_frexp_expf(d * (1.0f / 0.75f)) - 1;                     // vgetexp(d / 0.75)
  const float m = __builtin_amdgcn_ldexpf(d, -e);                                         // vgetmant: [0.75, 1.5)
  const vd_df x = df_div(df_add2_ff(-1.0f, m), df_add2_ff(1.0f, m));
  const vd_df x2 = df_squ(x);
  float t = 0.240320354700088500976562f;
  t = vd_fma(t, x2.x, 0.285112679004669189453125f);
  t = vd_fma(t, x2.x, 0.400007992982864379882812f);
  const vd_df c = {0.66666662693023681640625f, 3.69183861259614332084311e-09f};
  const vd_df l2 = {0.69314718246459960938f, -1.904654323148236017e-09f};
  vd_df s = df_mul_2f(l2, (float)e);
  const vd_df x_2 = {x.x * 2.0f, x.y * 2.0f};
  s = df_add_22(s, x_2);
  s = df_add_22(s, df_mul_22(df_mul_22(x2, x), df_add2_22(df_mul_2f(x2, t), c)));
  return s;
}
VD_DEV float vd_sleef_expkf(vd_df d) {
  float u = (d.x + d.y) * 1.442695040888963407359924681001892137426645954152985934135449406931f;
  const float qf = __builtin_rintf(u);
  vd_df s = df_add2_2f(d, qf * -0.693145751953125f);
  s = df_add2_2f(s, qf * -1.428606765330187045e-06f);
  s = df_norm(s);
  u = 0.00136324646882712841033936f;
  u = vd_fma(u, s.x, 0.00836596917361021041870117f);
  u = vd_fma(u, s.x, 0.0416710823774337768554688f);
  u = vd_fma(u, s.x, 0.166665524244308471679688f);
  u = vd_fma(u, s.x, 0.499999850988388061523438f);
  vd_df t = df_add2_22(s, df_mul_2f(df_squ(s), u));
  t = df_add_f2(1.0f, t);
  u = __builtin_amdgcn_ldexpf(t.x + t.y, (int)qf);
  return d.x < -104.0f ? 0.0f : u;
}
// 64 (intercept, slope) pairs of AVX-512 VRSQRT14 (index = exponent parity * 32 + top 5 mantissa bits), see vd_rsqrt14
static __constant__ const int2 c_vd_rs14[64] = {
{67102976,2002}, {65052928,1910}, {63096064,1830}, {61223424,1754}, {59428352,1682}, {57706240,1614}, {56052992,1550}, {54464768,1494},
{52935168,1438}, {51462400,1386}, {50042624,1338}, {48673792,1294}, {47350272,1250}, {46070272,1206}, {44834560,1170}, {43637504,1134},
{42477312,1098}, {41353984,1066}, {40263424,1034}, {39204864,1002}, {38178048,974}, {37180160,946}, {36210688,922}, {35267328,898},
{34348800,874}, {33454848,850}, {32585216,830}, {31735296,806}, {30908160,786}, {30103040,770}, {29314816,750}, {28547584,734},
{27792640,1414}, {26343680,1350}, {24960000,1294}, {23634944,1238}, {22367232,1190}, {21149440,1142}, {19980544,1098}, {18856192,1054},
{17775872,1018}, {16734976,982}, {15729920,946}, {14761216,914}, {13825280,882}, {12921344,854}, {12046592,826}, {11201280,802},
{10381056,778}, {9585408,754}, {8814336,730}, {8067328,710}, {7340800,690}, {6635008,670}, {5948416,650}, {5281792,634},
{4633088,618}, {4001024,602}, {3385088,586}, {2784768,570}, {2200832,558}, {1629440,542}, {1073152,530}, {529920,518}
};
VD_DEV float vd_rsqrt14(float x, const int2* __restrict__ tab) {   // x normal and > 0: the instruction's result, bit for bit
  const uint32_t b = __float_as_uint(x);
  const int e = (int)(b >> 23) - 127, odd = e & 1, half = (e - odd) / 2;
  const uint32_t m = b & 0x7fffffu;
  const int2 ab = tab[odd * 32 + (int)(m >> 18)];
  uint32_t r = 0x3f000000u + ((uint32_t)((ab.x - ab.y * (int)((m >> 8) & 0x3ffu)) >> 10) << 7);
  if (m == 0 && !odd) r = 0x3f800000u;                           // exact powers of 4 come back exactly
  return __uint_as_float(r - ((uint32_t)half << 23));
}
// torch.sqrt (MKL VML vsSqrt, high accuracy, AVX-512 path): y = VRSQRT14(x); s = x y; h = y / 2; fma(fma(-s, s, x), h, s).
// tab: c_vd_rs14 or a copy of it in LDS (vd_stage_rs14).
VD_DEV float vd_sqrt_torch(float x, const int2* __restrict__ tab) {
  if (!(x >= 0x1p-100f) || x == __builtin_inff()) return sqrtf(x);  // 0 (every flat pixel), < 2^-100 (not reproduced, see the oracle), inf, nan
  const float y = vd_rsqrt14(x, tab), s = x * y, h = 0.5f * y;
  return vd_fma(vd_fma(-s, s, x), h, s);
}
VD_DEV void vd_stage_rs14(int2* lds, int tid, int nthreads) {      // caller synchronises
  for (int i = tid; i < 64; i += nthreads) lds[i] = c_vd_rs14[i];
}
// torch.pow(x, e) for x >= 0, e a Python float: ATen's special exponents first (0 -> 1, 1 -> x, 0.5 -> sqrt, 2 -> x*x, 3 -> x*x*x,
// -0.5 / -1 / -2 the reciprocals), else SLEEF.  e is uniform over a launch, so the branches are.
VD_DEV float vd_pow_torch(float x, float e, const int2* __restrict__ rs14) {
  if (e == 0.0f) return 1.0f;
  if (e == 1.0f) return x;
  if (e == 0.5f) return vd_sqrt_torch(x, rs14);
  if (e == 2.0f) return x * x;
  if (e == 3.0f) return (x * x) * x;
  if (e == -0.5f) return 1.0f / sqrtf(x);
  if (e == -1.0f) return 1.0f / x;
  if (e == -2.0f) return 1.0f / (x * x);
  if (x == 0.0f) return e < 0.0f ? __builtin_inff() : 0.0f;
  if (x == 1.0f) return 1.0f;
  return vd_sleef_expkf(df_mul_2f(vd_sleef_logkf(x), e));
}
VD_DEV float vd_pow15_torch(float x) {                             // torch.pow(x, 1.5), x >= 0
  if (x == 0.0f) return 0.0f;
  if (x == 1.0f) return 1.0f;
  return vd_sleef_expkf(df_mul_2f(vd_sleef_logkf(x), 1.5f));
}
VD_DEV float vd_sleef_expf(float d) {
  const float qf = __builtin_rintf(d * 1.442695040888963407359924681001892137426645954152985934135449406931f);
  const int q = (int)qf;
  float s = vd_fma(qf, -0.693145751953125f, d);
  s = vd_fma(qf, -1.428606765330187045e-06f, s);
  float u = 0.000198527617612853646278381f;
  u = vd_fma(u, s, 0.00139304355252534151077271f);
  u = vd_fma(u, s, 0.00833336077630519866943359f);
  u = vd_fma(u, s, 0.0416664853692054748535156f);
  u = vd_fma(u, s, 0.166666671633720397949219f);
  u = vd_fma(u, s, 0.5f);
  u = 1.0f + vd_fma(s * s, u, s);
  u = __builtin_amdgcn_ldexpf(__builtin_amdgcn_ldexpf(u, q >> 1), q - (q >> 1));
  if (d < -104.0f) u = 0.0f;
  if (d > 100.0f) u = __builtin_inff();
  return u;
}
VD_DEV float vd_sigmoid_torch(float x) { return 1.0f / (1.0f + vd_sleef_expf(0.0f - x)); }

// ---- ATen's scalar tails (round 5; oracle/vd3d_oracle.c "ATen's SCALAR TAILS" has the derivation) --------------------------------------------
// torch.pow / torch.sigmoid on a contiguous n-element float32 CPU plane: min(threads, ceil(n / 32768)) chunks of ceil(n / that) elements (one chunk
// below 32768 elements or with one thread), and the last (chunk length mod 32) elements of every chunk go through libm instead of SLEEF.
// vd_tails_of runs on the host; kernels that evaluate pow / sigmoid are instantiated with and without the tail test, so planes without
// tails (every video size) run the code they always ran.
struct vd_tails { unsigned chunk, n; int on; };
static inline vd_tails vd_tails_of(unsigned long long n, int threads) {
  vd_tails t; t.chunk = n ? (unsigned)n : 1u; t.n = (unsigned)n; t.on = 0;
  if (threads <= 0 || n == 0 || n >= (1ull << 32)) return t;
  unsigned long long nt = (n + 32767) / 32768;
  if (nt > (unsigned long long)threads) nt = (unsigned long long)threads;
  if (n < 32768 || nt < 1) nt = 1;
  const unsigned long long chunk = (n + nt - 1) / nt, last = n - (n - 1) / chunk * chunk;
  t.chunk = (unsigned)chunk;
  t.on = ((n > chunk) && (chunk & 31)) || (last & 31) ? 1 : 0;
  return t;
}
static inline vd_tails vd_tails_all(unsigned long long n) { vd_tails t; t.chunk = 1u; t.n = (unsigned)n; t.on = 2; return t; }   // diagnostic: every element (vd3d_torch_math_aten)
VD_DEV bool vd_in_tail(const vd_tails& t, unsigned i) {
  if (t.on == 2) return true;
  const unsigned start = i / t.chunk * t.chunk, rem = t.n - start, len = rem < t.chunk ? rem : t.chunk;
  return i - start >= (len & ~31u);
}
// glibc 2.35 expf (sysdeps/ieee754/flt-32/e_expf.c, the FMA-contracted x86-64 build): N = 32 table of 2^(i/32) as double bit patterns minus i << 47,
// a cubic in double, one rounding to float32 at the end.  Checked on the CPU against libm on all 2^32 inputs (oracle), here against the oracle.
static __constant__ const unsigned long long c_vd_exp2f_t[32] = {
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull, 0x3fef72b83c7d517bull, 0x3fef54873168b9aaull,
    0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull, 0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
    0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull, 0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull,
    0x3feea11473eb0187ull, 0x3feea589994cce13ull, 0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull, 0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full,
    0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};
VD_DEV float vd_expf_glibc(float x) {
  const unsigned ux = __float_as_uint(x), abstop = (ux >> 20) & 0x7ffu;
  if (abstop >= 0x42bu) {                                         // |x| >= 88, inf, nan
    if (ux == 0xff800000u) return 0.0f;
    if (abstop >= 0x7f8u) return x + x;
    if (x > 0x1.62e42ep6f) return __builtin_inff();
    if (x < -0x1.9fe368p6f) return 0.0f;
  }
  const double xd = (double)x, invln2n = 0x1.71547652b82fep+0 * 32, shift = 0x1.8p+52;
  const double c0 = 0x1.c6af84b912394p-5 / 32 / 32 / 32, c1 = 0x1.ebfce50fac4f3p-3 / 32 / 32, c2 = 0x1.62e42ff0c52d6p-1 / 32;
  const double z = invln2n * xd;                                   // (the library is built with -ffp-contract=off: every operation below is one rounding)
  double kd = z + shift;
  const unsigned long long ki = (unsigned long long)__double_as_longlong(kd);
  kd = kd - shift;
  const double r = __builtin_fma(invln2n, xd, -kd);
  const double sc = __longlong_as_double((long long)(c_vd_exp2f_t[ki & 31] + (ki << 47)));
  const double q = __builtin_fma(c0, r, c1), r2 = r * r;
  double y = __builtin_fma(c2, r, 1.0);
  y = __builtin_fma(q, r2, y);
  return (float)(y * sc);
}
VD_DEV float vd_sigmoid_tail(float x) { return 1.0f / (1.0f + vd_expf_glibc(-x)); }
// (float) std::pow((double) x, e): ocml's double pow is within an ULP of a double like glibc's, so the two agree on the float32 rounding except
// when x^e lies within ~2^-52 (relative) of a float32 rounding boundary
VD_DEV float vd_pow_tail(float x, double e) { return (float)pow((double)x, e); }
static inline __host__ __device__ bool vd_pow_is_special(float e) { return e == 0.0f || e == 1.0f || e == 0.5f || e == 2.0f || e == 3.0f || e == -0.5f || e == -1.0f || e == -2.0f; }

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

// ATen's OTHER float32 bilinear kernel (cpu_upsample_linear_channels_last, taken for outputs with oh + ow <= 128 and for 3-channel inputs when torch runs one
// intra-op thread; round 5, oracle/vd3d_oracle.c vo_interp_bilinear_aten): row and column weights multiplied first, then ONE chain starting at the second tap of the
// top row.  Part of the N-thread ATen mode (vd3d_shift_params::aten_threads >= 1); without it the nested form above is used at every size.
VD_DEV float vd_bilerp_pm(float p00, float p01, float p10, float p11, float wx0, float wx1, float wy0, float wy1) {
  const float w00 = wy0 * wx0, w01 = wy0 * wx1, w10 = wy1 * wx0, w11 = wy1 * wx1;
  float t = p01 * w01;
  t = vd_fma(p00, w00, t);
  t = vd_fma(p10, w10, t);
  return vd_fma(p11, w11, t);
}
VD_DEV float vd_bilerp_sel(bool pm, float p00, float p01, float p10, float p11, float wx0, float wx1, float wy0, float wy1) {
  return pm ? vd_bilerp_pm(p00, p01, p10, p11, wx0, wx1, wy0, wy1) : vd_bilerp(p00, p01, p10, p11, wx0, wx1, wy0, wy1);
}
static inline __host__ __device__ bool vd_interp_premult(int C, int oh, int ow, int aten_threads) {
  return aten_threads >= 1 && (oh + ow <= 128 || (aten_threads == 1 && C == 3));
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
