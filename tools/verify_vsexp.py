"""Exhaustive check of the torch.exp (oneMKL VML vsExp, AVX-512 high-accuracy path) restatement against torch itself: every float32 whose
magnitude lies in [2^-40, 87], both signs, through the oracle's exp_torch AND the product's host routine (vd3d_debug_exp_torch).
Development container only (needs torch's CPU kernels; ~10 minutes on 8 threads).  python tools/verify_vsexp.py > profiles/r04_vsexp_sweep.md"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O  # noqa: E402
from visiondepth3d_amd import _lib  # noqa: E402


def main():
    L = _lib.lib()
    f2u = lambda v: int(np.float32(v).view(np.uint32))
    lo, hi = f2u(2.0 ** -40), f2u(87.0)
    print("# torch.exp (MKL vsExp) restatement vs torch, every float32 with |x| in [2^-40, 87]\n")
    print(f"torch {torch.__version__}, CPU capability {torch.backends.cpu.get_cpu_capability()}, MKL {torch.backends.mkl.is_available()}\n")
    print("| sign | inputs | oracle exp_torch != torch | host_exp_torch != torch | torch != rounded exp |")
    print("|---|---|---|---|---|")
    step = 1 << 26
    t0 = time.time()
    for neg in (True, False):
        n = bo = bh = bc = 0
        for a in range(lo, hi + 1, step):
            bits = np.arange(a, min(a + step, hi + 1), dtype=np.uint32)
            if neg:
                bits |= np.uint32(0x80000000)
            x = np.ascontiguousarray(bits.view(np.float32))
            t = torch.exp(torch.from_numpy(x)).numpy().view(np.uint32)
            bo += int(np.count_nonzero(O.torch_math("exp", x).view(np.uint32) != t))
            h = np.empty_like(x)
            L.vd3d_debug_exp_torch(x.ctypes.data_as(C.c_void_p), h.ctypes.data_as(C.c_void_p), x.size)
            bh += int(np.count_nonzero(h.view(np.uint32) != t))
            bc += int(np.count_nonzero(np.exp(x.astype(np.float64)).astype(np.float32).view(np.uint32) != t))
            n += x.size
        print(f"| {'-' if neg else '+'} | {n} | {bo} | {bh} | {bc} ({100.0 * bc / n:.2f} %) |", flush=True)
    print(f"\n{time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
