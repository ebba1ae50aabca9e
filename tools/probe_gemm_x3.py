"""Split-bf16 GEMM (vd3d_gemm_x3) on the depth net's linear shapes: error vs float64 beside torch's float32 GEMM, and time per call.
usage: python tools/probe_gemm_x3.py [M]      (default M = 16 frames x 2443 tokens = 39088)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from visiondepth3d_amd.render_3d import Renderer


def bench(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 39088
    mode = os.environ.get("X3_MODE", "bf16x3")
    print("mode", mode)
    R = Renderer(0)
    g = torch.Generator(device="cuda").manual_seed(1)
    tot_x3 = tot_f32 = 0.0
    shapes = ((768, 2304, False), (768, 768, False), (768, 3072, True), (3072, 768, False), (384, 1152, False), (1024, 4096, True))
    if len(sys.argv) > 2 and sys.argv[2] == "one":   # counter passes: one shape, a few launches
        shapes = ((768, 2304, False),)
    for (K, N, gelu) in shapes:
        x = torch.randn(M, K, device="cuda", generator=g) * 2.0
        w = torch.randn(N, K, device="cuda", generator=g) * 0.05
        b = torch.randn(N, device="cuda", generator=g)
        img = R.gemm_x3_pack(w, mode)
        y = R.linear_x3(x, img, N, b, gelu=gelu, mode=mode)
        torch.cuda.synchronize()
        ms = 2048
        ref64 = torch.nn.functional.linear(x[:ms].double(), w.double(), b.double())
        if gelu:
            ref64 = torch.nn.functional.gelu(ref64)
        y32 = torch.nn.functional.linear(x[:ms], w, b)
        if gelu:
            y32 = torch.nn.functional.gelu(y32)
        scale = (x[:ms].abs().double() @ w.abs().double().T) + b.abs().double()
        e3 = ((y[:ms].double() - ref64).abs() / scale).max().item()
        e32 = ((y32.double() - ref64).abs() / scale).max().item()
        r3 = ((y[:ms].double() - ref64).pow(2).mean().sqrt() / ref64.pow(2).mean().sqrt()).item()
        r32 = ((y32.double() - ref64).pow(2).mean().sqrt() / ref64.pow(2).mean().sqrt()).item()
        # tail rows: the last tile is partial when M is not a multiple of 256
        tail = torch.nn.functional.linear(x[-300:].double(), w.double(), b.double())
        if gelu:
            tail = torch.nn.functional.gelu(tail)
        et = ((y[-300:].double() - tail).abs() / ((x[-300:].abs().double() @ w.abs().double().T) + b.abs().double())).max().item()
        t3 = bench(lambda: R.linear_x3(x, img, N, b, gelu=gelu, mode=mode))
        if gelu:
            t32 = bench(lambda: torch.nn.functional.gelu(torch.nn.functional.linear(x, w, b)))
        else:
            t32 = bench(lambda: torch.nn.functional.linear(x, w, b))
        fl = 2.0 * M * K * N
        print(f"M {M} K {K} N {N} gelu {int(gelu)}: x3 {t3:.3f} ms = {fl / t3 / 1e9:.0f} TF-equiv ({6 * fl / t3 / 1e9:.0f} TF bf16 MFMA) | torch f32 {t32:.3f} ms = {fl / t32 / 1e9:.0f} TF"
              f" | max err / sum|x||w|: x3 {e3:.2e} (tail rows {et:.2e}) f32 {e32:.2e} | rel rms: x3 {r3:.2e} f32 {r32:.2e}", flush=True)
        if K == 768 or K == 3072:
            tot_x3 += t3; tot_f32 += t32
    print(f"one DA-V2-Base layer (qkv + proj + fc1[+gelu] + fc2): x3 {tot_x3:.3f} ms, torch f32 {tot_f32:.3f} ms; x 12 layers: {12 * tot_x3:.1f} vs {12 * tot_f32:.1f} ms")


if __name__ == "__main__":
    main()
