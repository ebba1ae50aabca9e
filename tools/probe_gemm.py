#!/usr/bin/env python3
"""GPU probe: the four backbone GEMM shapes of DA-V2-Small at batch 16 (M = 40960), default hipBLASLt pick vs TunableOp."""
import os, time, torch, torch.nn.functional as F
M = 40960
shapes = {"qkv": (384, 1152), "proj": (384, 384), "fc1": (384, 1536), "fc2": (1536, 384)}
def bench(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
tot = 0
for name, (K, N) in shapes.items():
    x = torch.randn(M, K, device="cuda", dtype=torch.bfloat16); w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16); b = torch.randn(N, device="cuda", dtype=torch.bfloat16)
    us = bench(lambda: F.linear(x, w, b)); tot += us
    print(f"{name:5s} M={M} K={K} N={N}: {us:7.1f} us  {2*M*K*N/us/1e6:6.1f} TFLOP/s")
print(f"sum {tot:.1f} us per layer (x12 = {12*tot/1e3:.2f} ms), tunable={os.environ.get('PYTORCH_TUNABLEOP_ENABLED')}")
