/* Exhaustive check of the 3-operation exact division for the DOF blur weight of E1 (core/render_3d.py:793):
 *     |depth - focal| / (focus_width + 1e-6),  focus_width = 0.35 (the only value render_sbs_3d passes, :1357-1360)
 *     q0 = x * rc;  r = fma(-q0, d, x);  q = fma(r, rc, q0)        with d = (float)(0.35 + 1e-6), rc = RN(1/d)
 * against IEEE x / d for EVERY float x in [0, 1] (depth and focal are in [0, 1]); subnormal x included.
 * Build / run:  gcc -O2 -mfma -fopenmp tools/verify_fastdiv_fw.c -o /tmp/vfw && /tmp/vfw      (result recorded in vd3d_finish.hip) */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

int main(void) {
  const float d = (float)(0.35 + 1e-6), rc = 1.0f / d;
  const float one = 1.0f;
  uint32_t top; memcpy(&top, &one, 4);
  long long bad = 0;
#pragma omp parallel for reduction(+ : bad) schedule(static)
  for (long long b = 0; b <= (long long)top; ++b) {
    uint32_t u = (uint32_t)b; float x; memcpy(&x, &u, 4);
    const float q0 = x * rc;
    const float r = fmaf(-q0, d, x);
    const float q = fmaf(r, rc, q0);
    if (q != x / d) ++bad;
  }
  uint32_t db; memcpy(&db, &d, 4);
  printf("d=%.9g (0x%08x) rc=%.9g mismatches=%lld of %u\n", d, db, rc, bad, top + 1);
  return bad != 0;
}
