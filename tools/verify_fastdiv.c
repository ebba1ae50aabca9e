/* Exhaustive check of the 3-operation exact division used by k_warp_fused for the avg_pool2d normalisation
 *     q0 = x * rc;  r = fma(-q0, d, x);  q = fma(r, rc, q0)        with rc = RN(1/d), d = k*k
 * against IEEE x / d for EVERY float x in [0, d] (the window sums of k*k values in [0,1]) and every blur_ksize k = 1..33.
 * Build / run (development aid; result recorded in vd3d_warp.hip):  gcc -O2 -mfma -fopenmp tools/verify_fastdiv.c -o /tmp/vf/vf && /tmp/vf/vf
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

int main(void) {
  for (int k = 1; k <= 33; ++k) {
    const float d = (float)(k * k), rc = 1.0f / d;
    uint32_t top; memcpy(&top, &d, 4);
    long long bad = 0;
#pragma omp parallel for reduction(+ : bad) schedule(static)
    for (long long b = 0; b <= (long long)top; ++b) {
      uint32_t u = (uint32_t)b; float x; memcpy(&x, &u, 4);
      const float q0 = x * rc;
      const float r = fmaf(-q0, d, x);
      const float q = fmaf(r, rc, q0);
      const float t = x / d;
      if (q != t) ++bad;
    }
    printf("k=%2d d=%4.0f mismatches=%lld of %u\n", k, d, bad, top + 1);
    fflush(stdout);
  }
  return 0;
}
