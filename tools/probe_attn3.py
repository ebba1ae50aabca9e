#!/usr/bin/env python3
"""GPU probe: does SDPA's output follow a dense-permuted query layout (so that the [B,nh,T,hd] -> [B,T,d] reshape is free)?"""
import time, torch, torch.nn.functional as F
B, nh, T, hd, Tk = 16, 6, 2560, 64, 2443
d = nh * hd
h = torch.randn(B * T, d, device="cuda", dtype=torch.bfloat16)
w3 = torch.randn(3, d, d, device="cuda", dtype=torch.bfloat16) * 0.05
b3 = torch.randn(3, 1, d, device="cuda", dtype=torch.bfloat16)
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
qkv3 = torch.baddbmm(b3, h.unsqueeze(0).expand(3, B * T, d), w3.transpose(1, 2))      # [3, B*T, d], each slice dense
q = qkv3[0].view(B, T, nh, hd).transpose(1, 2)
k = qkv3[1].view(B, T, nh, hd)[:, :Tk].transpose(1, 2)
v = qkv3[2].view(B, T, nh, hd)[:, :Tk].transpose(1, 2)
o = F.scaled_dot_product_attention(q, k, v)
print("q strides", q.stride(), "o strides", o.stride(), "o.transpose(1,2) contiguous:", o.transpose(1, 2).is_contiguous())
wq = torch.cat([w3[0], w3[1], w3[2]], 0).contiguous(); bq = b3.reshape(-1)
def fused():
    qkv = F.linear(h, wq, bq).view(B, T, 3, nh, hd)
    qq, kk, vv = qkv[:, :, 0].transpose(1, 2), qkv[:, :Tk, 1].transpose(1, 2), qkv[:, :Tk, 2].transpose(1, 2)
    return F.scaled_dot_product_attention(qq, kk, vv).transpose(1, 2).reshape(B, T, d)
def batched():
    x3 = torch.baddbmm(b3, h.unsqueeze(0).expand(3, B * T, d), w3.transpose(1, 2))
    qq = x3[0].view(B, T, nh, hd).transpose(1, 2)
    kk = x3[1].view(B, T, nh, hd)[:, :Tk].transpose(1, 2); vv = x3[2].view(B, T, nh, hd)[:, :Tk].transpose(1, 2)
    return F.scaled_dot_product_attention(qq, kk, vv).transpose(1, 2).reshape(B, T, d)
wqo, bqo = w3[0].contiguous(), b3[0].reshape(-1).contiguous()
wkv, bkv = torch.cat([w3[1], w3[2]], 0).contiguous(), torch.cat([b3[1].reshape(-1), b3[2].reshape(-1)]).contiguous()
def split_q():
    qq = F.linear(h, wqo, bqo).view(B, T, nh, hd).transpose(1, 2)                       # dense-permuted: the output follows it
    kv = F.linear(h, wkv, bkv).view(B, T, 2, nh, hd)
    kk, vv = kv[:, :Tk, 0].transpose(1, 2), kv[:, :Tk, 1].transpose(1, 2)
    return F.scaled_dot_product_attention(qq, kk, vv).transpose(1, 2).reshape(B, T, d)
print(f"q GEMM + fused kv GEMM + sdpa + free reshape: {bench(split_q):7.1f} us   diff {(fused().float() - split_q().float()).abs().max().item()}")
print(f"fused-linear qkv + sdpa + reshape: {bench(fused):7.1f} us")
print(f"batched qkv (dense slices) + sdpa + reshape: {bench(batched):7.1f} us")
print("max abs diff", (fused().float() - batched().float()).abs().max().item())
