#!/usr/bin/env python3
"""Offline pinning of the oracle's torch-CPU numerics (pow_torch / sigmoid_torch / sqrt_torch, oracle/vd3d_oracle.c) against torch
itself, far denser than tests/test_torch_cpu_numerics.py, and re-derivation of the VRSQRT14 table from the instruction.

    python tools/verify_sleef.py            # ~10 min on 8 cores: the sweeps quoted in the oracle header
    python tools/verify_sleef.py --quick    # every 64th float of the same ranges
    python tools/verify_sleef.py --table    # re-derive RS14[64][2] from _mm_rsqrt14_ss (needs an AVX-512 CPU and gcc) and print it

Needs a torch whose CPU kernels are the AVX-512 ones (torch.backends.cpu.get_cpu_capability() == "AVX512") with MKL: that is the
build the reference fixtures were generated with.  One torch thread and arrays of a multiple of 32 elements: no scalar tails."""
import argparse
import os
import subprocess
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def sweep(name, fn_torch, op, param, lo_bits, hi_bits, step, O, torch):
    bad = tot = 0
    first = None
    CH = 1 << 25
    for a in range(lo_bits, hi_bits, CH * step):
        bits = np.arange(a, min(hi_bits, a + CH * step), step, dtype=np.uint32)
        bits = bits[: bits.size // 32 * 32]
        if not bits.size:
            continue
        x = bits.view(np.float32).copy()
        t = fn_torch(torch.from_numpy(x)).numpy()
        m = O.torch_math(op, x, param)
        ne = np.nonzero((t.view(np.uint32) != m.view(np.uint32)) & ~(np.isnan(t) & np.isnan(m)))[0]
        bad += ne.size
        tot += x.size
        if ne.size and first is None:
            i = ne[0]
            first = (float(x[i]).hex(), float(t[i]).hex(), float(m[i]).hex())
    print(f"{name}: {bad} mismatches of {tot}" + (f"  first (x, torch, oracle) = {first}" if first else ""), flush=True)
    return bad


def f2b(v):
    return int(np.float32(v).view(np.uint32))


RS_C = r"""
#include <immintrin.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
int main(void) {                       /* VRSQRT14 of the 2^16 inputs that matter: 15 mantissa bits x exponent parity, x in [1, 4) */
  for (uint32_t i = 0; i < (1u << 16); ++i) {
    uint32_t b = 0x3f800000u + (i << 8), o; float x, y; memcpy(&x, &b, 4);
    __m128 v = _mm_set_ss(x); y = _mm_cvtss_f32(_mm_rsqrt14_ss(v, v)); memcpy(&o, &y, 4);
    /* the low 8 mantissa bits are ignored by the instruction: check one more point per step */
    uint32_t b2 = b + 255, o2; memcpy(&x, &b2, 4); v = _mm_set_ss(x); y = _mm_cvtss_f32(_mm_rsqrt14_ss(v, v)); memcpy(&o2, &y, 4);
    if (o2 != o && i != 0) { printf("LOWBITS\n"); return 1; }     /* i == 0: x = 1.0 itself is returned exactly */
    printf("%u\n", o);
  }
  return 0;
}
"""


def derive_table():
    with tempfile.TemporaryDirectory() as td:
        src, exe = os.path.join(td, "rs.c"), os.path.join(td, "rs")
        open(src, "w").write(RS_C)
        subprocess.run(["gcc", "-O2", "-mavx512f", "-mavx512vl", "-o", exe, src], check=True)
        out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split()
    o = np.array([int(v) for v in out], dtype=np.int64)
    v = (o - 0x3F000000) >> 7                       # 16-bit mantissa field of results in (0.5, 1]
    K, tab = 1024, []
    for j in range(64):                             # index = parity * 32 + top 5 mantissa bits; 10 interpolation bits
        k, b = np.arange(K), v[j * K:(j + 1) * K]
        if j == 0:
            k, b = k[1:], b[1:]                     # x = 1.0 returns exactly 1.0 (special case in rsqrt14())
        est, sol = (b[0] - b[-1]) / (k[-1] - k[0]), None
        for B in range(int((est - 0.05) * 1024), int((est + 0.05) * 1024) + 2):
            lo, hi = (b * 1024 + B * k).max(), ((b + 1) * 1024 + B * k).min()
            if lo < hi:
                sol = (int(lo), B)
                break
        assert sol, j
        assert np.array_equal((sol[0] - sol[1] * k) >> 10, b), j
        tab.append(sol)
    print(",\n".join(", ".join("{%d,%d}" % t for t in tab[i:i + 8]) for i in range(0, 64, 8)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--table", action="store_true")
    a = ap.parse_args()
    if a.table:
        return derive_table()
    import torch
    from oracle import oracle as O
    O.build()
    assert torch.backends.cpu.get_cpu_capability() == "AVX512" and torch.backends.mkl.is_available(), "different torch CPU code paths"
    torch.set_num_threads(1)
    q = 64 if a.quick else 1
    bad = 0
    bad += sweep("pow 0.85, every float32 of [2^-40, 1]", lambda t: torch.pow(t, 0.85), "pow", 0.85, f2b(2.0 ** -40), f2b(1.0) + 32, q, O, torch)
    bad += sweep("pow 1.5, every float32 of [2^-80, 1] (normal results)", lambda t: torch.pow(t, 1.5), "pow", 1.5, f2b(2.0 ** -80), f2b(1.0) + 32, q, O, torch)
    for g in (0.7, 0.75, 0.9, 0.999, 1.1, 1.2, 1.3, 2.2):
        bad += sweep(f"pow {g}, every {37 * q}th float32 of [2^-40, 1]", lambda t: torch.pow(t, g), "pow", g, f2b(2.0 ** -40), f2b(1.0) + 32, 37 * q, O, torch)
    for sgn in (0, 0x80000000):
        bad += sweep("sigmoid, every %dth float32 of %s[2^-30, 110]" % (7 * q, "-" if sgn else ""), torch.sigmoid, "sigmoid", 0.0,
                     sgn + f2b(2.0 ** -30), sgn + f2b(110.0), 7 * q, O, torch)
    bad += sweep("sqrt, every float32 of [1, 4)", torch.sqrt, "sqrt", 0.0, f2b(1.0), f2b(4.0), q, O, torch)
    bad += sweep(f"sqrt, every {61 * q}th float32 of [2^-100, 2^127]", torch.sqrt, "sqrt", 0.0, f2b(2.0 ** -100), f2b(2.0 ** 127), 61 * q, O, torch)
    print("TOTAL mismatches:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
