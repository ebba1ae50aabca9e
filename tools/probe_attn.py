#!/usr/bin/env python3
"""GPU probe: SDPA backends for the DINOv2 attention shape of the 1080p workload (development aid)."""
import time, torch, torch.nn.functional as F
from torch.nn.attention import sdpa_kernel, SDPBackend
B, nh, T, hd = 16, 6, 2443, 64
q, k, v = (torch.randn(B, T, 3, nh, hd, device="cuda", dtype=torch.bfloat16)[:, :, i].transpose(1, 2) for i in range(3))
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
fl = 4.0 * B * nh * T * T * hd
for name, be in (("flash", SDPBackend.FLASH_ATTENTION), ("efficient", SDPBackend.EFFICIENT_ATTENTION), ("math", SDPBackend.MATH)):
    try:
        with sdpa_kernel(be):
            us = bench(lambda: F.scaled_dot_product_attention(q, k, v))
        print(f"{name:10s} {us:8.1f} us  {fl/us/1e6:7.1f} TFLOP/s")
    except Exception as e:
        print(name, "failed:", str(e)[:100])
print("preferred_rocm_fa_library:", getattr(torch.backends.cuda, "preferred_rocm_fa_library", None))
try:
    torch.backends.cuda.preferred_rocm_fa_library("ck")
    with sdpa_kernel(SDPBackend.FLASH_ATTENTION):
        us = bench(lambda: F.scaled_dot_product_attention(q, k, v))
    print(f"flash(ck)  {us:8.1f} us  {fl/us/1e6:7.1f} TFLOP/s")
except Exception as e:
    print("ck failed:", str(e)[:200])
# contiguous [B,nh,T,hd] layout instead of the strided qkv view
qc, kc, vc = q.contiguous(), k.contiguous(), v.contiguous()
us = bench(lambda: F.scaled_dot_product_attention(qc, kc, vc)); print(f"contig     {us:8.1f} us")
# padded sequence (multiple of 128) with the padding masked out would need a mask; time the unmasked padded shape as a bound
Tp = 2560
qp = torch.randn(B, nh, Tp, hd, device="cuda", dtype=torch.bfloat16)
us = bench(lambda: F.scaled_dot_product_attention(qp, qp, qp)); print(f"T=2560     {us:8.1f} us  (unmasked bound)")
Tp = 2432
qp = torch.randn(B, nh, Tp, hd, device="cuda", dtype=torch.bfloat16)
us = bench(lambda: F.scaled_dot_product_attention(qp, qp, qp)); print(f"T=2432     {us:8.1f} us")
