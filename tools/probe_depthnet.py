#!/usr/bin/env python3
"""GPU probe: depth-net forward time under dtype / memory-format / MIOpen-find variants (development aid)."""
import itertools, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visiondepth3d_amd.depth import DepthPipe

def bench(pipe, B=8, iters=5):
    x = torch.randint(0, 255, (B, 1080, 1920, 3), dtype=torch.uint8, device="cuda")
    for _ in range(2):
        pipe.infer_bgr_u8(x)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(iters):
        pipe.infer_bgr_u8(x)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / iters * 1e3

name = sys.argv[1] if len(sys.argv) > 1 else "depth-anything-v2-small"
for dtype, cl, bm in itertools.product((torch.bfloat16, torch.float16), (True, False), (False, True)):
    torch.backends.cudnn.benchmark = bm
    try:
        p = DepthPipe(name, dtype=dtype, channels_last=cl)
        ms = bench(p)
        print(f"{name} dtype={dtype} channels_last={cl} cudnn.benchmark={bm}: {ms:.2f} ms / batch of 8  ({ms/8:.2f} ms/frame)", flush=True)
    except Exception as e:
        print("FAILED", dtype, cl, bm, repr(e)[:200], flush=True)
