#!/usr/bin/env python3
"""GPU tool: hipBLASLt / rocBLAS solution selection for the depth net's float32 GEMM shapes through PyTorch's TunableOp.

    python tools/tune_gemm.py tune  <csv>     # tune every shape (writes <csv>; PyTorch adds the device ordinal to the name)
    python tools/tune_gemm.py bench [<csv>]   # time the shapes with the default pick (no csv) or with the tuned table

Shapes: DA-V2-Base / -Small backbone at 518x924 (T = 2443 tokens), batches of 16 and 8 frames; float32 (the reference's precision)."""
import os, sys, time, glob
mode = sys.argv[1]
csv = sys.argv[2] if len(sys.argv) > 2 else None
if mode == "tune":
    os.environ["PYTORCH_TUNABLEOP_ENABLED"] = "1"
    os.environ["PYTORCH_TUNABLEOP_TUNING"] = "1"
    os.environ["PYTORCH_TUNABLEOP_FILENAME"] = csv
    os.environ.setdefault("PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS", "40")
    os.environ.setdefault("PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS", "5")
elif csv:
    os.environ["PYTORCH_TUNABLEOP_ENABLED"] = "1"
    os.environ["PYTORCH_TUNABLEOP_TUNING"] = "0"
    os.environ["PYTORCH_TUNABLEOP_FILENAME"] = csv
import torch, torch.nn.functional as F

T = 2443
def shapes():
    for d in (768, 384):
        for B in (16, 8):
            M = B * T
            for n_out, k_in, nm in ((3 * d, d, "qkv"), (d, d, "proj"), (4 * d, d, "fc1"), (d, 4 * d, "fc2")):
                yield d, B, M, k_in, n_out, nm

def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

tot = {}
for d, B, M, K, N, nm in shapes():
    x = torch.randn(B, T, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.02; b = torch.randn(N, device="cuda")
    ms = bench(lambda: F.linear(x, w, b))
    tot[(d, B)] = tot.get((d, B), 0.0) + ms
    print(f"{mode:5s} d={d} B={B} {nm:4s} [{M}x{K}]x[{K}x{N}]: {ms:7.3f} ms {2.0 * M * K * N / ms / 1e9:7.1f} TFLOP/s", flush=True)
for k, v in tot.items():
    print(f"{mode:5s} d={k[0]} B={k[1]}: {v:.3f} ms per layer, {12 * v:.2f} ms per 12 layers")
if mode == "tune":
    torch.cuda.tunable.write_file()
    print("written:", glob.glob(os.path.splitext(csv)[0] + "*"))
