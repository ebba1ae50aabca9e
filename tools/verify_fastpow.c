/* Exhaustive check of the table-driven float64 pow used by k_chain_shape (vd_pow_fast in vd3d_dev.h, restated here with the
 * SAME float64 operations: fma / mul / add / rint / ldexp, no contraction) against (float)pow((double)x, g):
 *   for every float x in (0, 1] the fast value either equals the reference rounding or is flagged "ambiguous" (result within
 *   2^-40 relative of a float32 rounding boundary), in which case the kernel falls back to the libm pow.
 * Prints, per exponent g, the number of unflagged mismatches (must be 0) and the fallback rate.
 * Build / run (development aid): gcc -O2 -mfma -ffp-contract=off -fopenmp tools/verify_fastpow.c -o /tmp/vf/fp -lm && /tmp/vf/fp 0.85
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../visiondepth3d_amd/csrc/vd3d_pow_tables.h"
#define X_NONE(v)
#define X_VAL(v) v,
static const double T_INVC[128] = {VD_POW_TABLES(X_VAL, X_NONE, X_NONE)};
static const double T_LOGC[128] = {VD_POW_TABLES(X_NONE, X_VAL, X_NONE)};
static const double T_EXP[64] = {VD_POW_TABLES(X_NONE, X_NONE, X_VAL)};
static void build_tables(void) {}
/* returns 1 and *out when the result is unambiguous, 0 when the caller must use the exact pow */
static int pow_fast(float x, double g, float* out) {
  uint32_t b; memcpy(&b, &x, 4);
  if (b < 0x00800000u || b > 0x3f800000u) return 0;          /* zero, subnormal, > 1: exact path */
  const int e = (int)(b >> 23) - 127;
  const int i = (b >> 16) & 0x7f;
  uint32_t mb = (b & 0x007fffffu) | 0x3f800000u; float mf; memcpy(&mf, &mb, 4);
  const double m = (double)mf;
  const double r = fma(m, T_INVC[i], -1.0);                     /* |r| <= 2^-8 */
  /* log2(1 + r) = r * (C1 + r*(C2 + r*(C3 + r*(C4 + r*(C5 + r*C6))))) , Ck = (-1)^(k+1) / (k ln 2) */
  const double C1 = 1.4426950408889634074, C2 = -0.72134752044448170368, C3 = 0.48089834696298780245, C4 = -0.36067376022224085184,
               C5 = 0.28853900817779268147, C6 = -0.24044917348149390123, C7 = 0.20609929155556620106;
  double p = fma(r, C7, C6); p = fma(r, p, C5); p = fma(r, p, C4); p = fma(r, p, C3); p = fma(r, p, C2); p = fma(r, p, C1);
  const double L = ((double)e + T_LOGC[i]) + r * p;
  const double y = g * L;                                          /* <= 0 */
  if (!(y > -120.0)) return 0;
  const double k = rint(y * 64.0);
  const double f = fma(k, -1.0 / 64.0, y);                         /* |f| <= 2^-7 */
  const long long ki = (long long)k;
  const int j = (int)(ki & 63);
  const int n = (int)((ki - j) / 64);
  const double t = f * 0.69314718055994530942;                     /* f ln 2, |t| <= 2^-7.5 */
  /* e^t = 1 + t (1 + t/2 (1 + t/3 (1 + t/4 (1 + t/5 (1 + t/6))))) */
  double q = fma(t, 1.0 / 6.0, 1.0); q = fma(t * (1.0 / 5.0), q, 1.0); q = fma(t * (1.0 / 4.0), q, 1.0);
  q = fma(t * (1.0 / 3.0), q, 1.0); q = fma(t * 0.5, q, 1.0); q = fma(t, q, 1.0);
  const double res = ldexp(T_EXP[j] * q, n);
  const float fr = (float)res;
  /* ambiguity test: distance of res from the nearest float32 rounding boundary, relative to res */
  uint64_t rb; memcpy(&rb, &res, 8);
  const uint32_t low = (uint32_t)(rb & 0x1fffffffu);               /* the 29 mantissa bits below float32 precision */
  const uint32_t half = 0x10000000u;
  const uint32_t dist = low > half ? low - half : half - low;      /* in units of 2^-52 of the mantissa */
  if (dist < (1u << 13)) return 0;                                 /* within 2^-39 relative of a boundary */
  *out = fr;
  return 1;
}
int main(int argc, char** argv) {
  build_tables();
  for (int a = 1; a < argc || a == 1; ++a) {
    const double g = a < argc ? atof(argv[a]) : 0.85;
    long long bad = 0, amb = 0, tot = 0;
#pragma omp parallel for reduction(+ : bad, amb, tot) schedule(static)
    for (long long bi = 1; bi <= 0x3f800000ll; ++bi) {
      uint32_t u = (uint32_t)bi; float x; memcpy(&x, &u, 4);
      float fv;
      ++tot;
      if (!pow_fast(x, g, &fv)) { ++amb; continue; }
      const float ref = (float)pow((double)x, g);
      if (fv != ref) ++bad;
    }
    printf("g=%g: unflagged mismatches=%lld, fallbacks=%lld of %lld (%.3g)\n", g, bad, amb, tot, (double)amb / (double)tot);
    if (a >= argc) break;
  }
  return 0;
}
