#!/usr/bin/env python3
"""GPU probe (development aid): where does the float32 depth net spend its time?  Per-op timings at the DA-V2 shapes
(B = 16, T = 2443 tokens) for the GEMMs, the attention variants (SDPA backends, query padding, explicit two-GEMM softmax)
and a torch-profiler kernel table of one whole forward."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F


def bench(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def attn_probe(d, nh, dt):
    B, T = 16, 2443
    hd = d // nh
    for Tq in (T, 2560):
        q = torch.randn(B, nh, Tq, hd, device="cuda", dtype=dt)
        k = torch.randn(B, nh, T, hd, device="cuda", dtype=dt)
        v = torch.randn(B, nh, T, hd, device="cuda", dtype=dt)
        fl = 4.0 * B * nh * Tq * T * hd
        for name, be in (("default", None), ("efficient", torch.nn.attention.SDPBackend.EFFICIENT_ATTENTION),
                         ("flash", torch.nn.attention.SDPBackend.FLASH_ATTENTION), ("math", torch.nn.attention.SDPBackend.MATH)):
            try:
                if be is None:
                    ms = bench(lambda: F.scaled_dot_product_attention(q, k, v))
                else:
                    with torch.nn.attention.sdpa_kernel(be):
                        ms = bench(lambda: F.scaled_dot_product_attention(q, k, v))
                print(f"  sdpa[{name:9s}] d={d} {dt} Tq={Tq}: {ms:7.3f} ms  {fl / ms / 1e9:7.1f} TFLOP/s", flush=True)
            except Exception as e:
                print(f"  sdpa[{name:9s}] d={d} {dt} Tq={Tq}: FAILED {repr(e)[:100]}", flush=True)

        def two_gemm():
            s = torch.matmul(q, k.transpose(-1, -2)) * (hd ** -0.5)
            return torch.matmul(torch.softmax(s, -1), v)
        try:
            print(f"  two-GEMM softmax     d={d} {dt} Tq={Tq}: {bench(two_gemm, 5):7.3f} ms", flush=True)
        except Exception as e:
            print("  two-GEMM FAILED", repr(e)[:100])


def gemm_probe(d, dt):
    M = 16 * 2443
    x = torch.randn(M, d, device="cuda", dtype=dt)
    for n_out, k_in, nm in ((3 * d, d, "qkv"), (d, d, "proj"), (4 * d, d, "fc1"), (d, 4 * d, "fc2")):
        w = torch.randn(n_out, k_in, device="cuda", dtype=dt) * 0.02
        b = torch.randn(n_out, device="cuda", dtype=dt)
        xi = x if k_in == d else torch.randn(M, k_in, device="cuda", dtype=dt)
        ms = bench(lambda: F.linear(xi, w, b))
        print(f"  linear {nm:5s} [{M}x{k_in}]x[{k_in}x{n_out}] {dt}: {ms:7.3f} ms  {2.0 * M * k_in * n_out / ms / 1e9:7.1f} TFLOP/s", flush=True)


def forward_profile(name, dt, H, W):
    from visiondepth3d_amd.depth import DepthPipe
    from visiondepth3d_amd.render_3d import Renderer
    r = Renderer(0)
    pipe = DepthPipe(name, device="cuda", dtype=dt, renderer=r)
    x = torch.randint(0, 255, (16, H, W, 3), dtype=torch.uint8, device="cuda")
    ms = bench(lambda: pipe.infer_bgr_u8(x, raw=True), 5)
    fl = pipe.flops_per_frame(H, W) * 16
    print(f"{name} {dt} {W}x{H}: {ms:.2f} ms / 16 frames = {ms / 16:.3f} ms/frame, {fl / ms / 1e9:.1f} TFLOP/s of {fl / 16e9:.0f} GFLOP/frame", flush=True)
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        pipe.infer_bgr_u8(x, raw=True)
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=22, max_name_column_width=70), flush=True)
    del pipe
    r.close()


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all", "ops"):
        for d, nh in ((768, 12), (384, 6)):
            print(f"== d={d}")
            gemm_probe(d, torch.float32)
            attn_probe(d, nh, torch.float32)
        attn_probe(768, 12, torch.bfloat16)
    if which in ("all", "fwd"):
        forward_profile("depth-anything-v2-base", torch.float32, 2160, 3840)
        forward_profile("depth-anything-v2-small", torch.float32, 1080, 1920)
