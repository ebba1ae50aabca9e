#!/usr/bin/env python3
"""GPU probe: which sequence length (q or k) triggers the slow AOTriton path, and the flash op's logsumexp convention."""
import time, math, torch, torch.nn.functional as F
B, nh, T, hd = 16, 6, 2443, 64
torch.manual_seed(0)
qkv = torch.randn(B, T, 3, nh, hd, device="cuda", dtype=torch.bfloat16)
q, k, v = (qkv[:, :, i].transpose(1, 2) for i in range(3))
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
def pad(x, n):
    return F.pad(x, (0, 0, 0, n - x.shape[2]))
for tq, tk in ((2443, 2443), (2443, 2432), (2560, 2443), (2560, 2432), (2560, 2560), (2448, 2448), (2496, 2496)):
    qq = pad(q, tq) if tq >= T else q[:, :, :tq]
    kk = pad(k, tk) if tk >= T else k[:, :, :tk]
    vv = pad(v, tk) if tk >= T else v[:, :, :tk]
    print(f"Tq={tq} Tk={tk}: {bench(lambda: F.scaled_dot_product_attention(qq, kk, vv)):7.1f} us")
scale = 1.0 / math.sqrt(hd)
res = torch.ops.aten._scaled_dot_product_flash_attention(pad(q, 2560), pad(k, 2560), pad(v, 2560), 0.0, False, False, scale=scale)
out, lse = res[0], res[1]
print("out", out.shape, out.dtype, "lse", lse.shape, lse.dtype)
ref = F.scaled_dot_product_attention(q.float(), k.float(), v.float())
s = (q.float() @ k.float().transpose(-1, -2)) * scale
lse_ref = torch.logsumexp(torch.cat([s, torch.zeros(B, nh, T, 2560 - T, device="cuda")], -1), -1)
print("lse vs natural-log reference: max abs diff", (lse[:, :, :T] - lse_ref).abs().max().item())
corr = 1.0 / (1.0 - (2560 - T) * torch.exp(-lse[:, :, :T]))
fixed = out[:, :, :T].float() * corr[..., None]
print("zero-pad + LSE correction vs exact: max abs err", (fixed - ref).abs().max().item(), " plain bf16 sdpa err", (F.scaled_dot_product_attention(q, k, v).float() - ref).abs().max().item())
print("min D_real/(D_real+npad):", (1.0 / corr).min().item())
