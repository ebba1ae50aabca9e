"""Split-bf16 attention (vd3d_attention_x3) on the depth net's shapes: error vs float64 beside SDPA float32, time per call.
usage: python tools/probe_attn_x3.py [B]   (default 16 frames; DA-V2-Base at 4K: T = 2443, H = 12; at 1080p / Small: T = 1370, H = 6)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from visiondepth3d_amd.render_3d import Renderer
F = torch.nn.functional


def bench(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    mode = os.environ.get("X3_MODE", "bf16x3")
    print("mode", mode)
    R = Renderer(0)
    g = torch.Generator(device="cuda").manual_seed(1)
    for (T, H) in ((2443, 12), (1370, 6), (1370, 12), (2443, 12), (1370, 12)):
        D = 64
        qkv = torch.randn(B, T, 3, H, D, device="cuda", generator=g)
        scale = D ** -0.5
        flat = qkv.view(B, T, 3 * H * D)
        out = R.attention_x3(flat, H, scale, mode=mode)
        q, k, v = (qkv[:, :, i].transpose(1, 2) for i in range(3))
        o32 = F.scaled_dot_product_attention(q, k, v, scale=scale).transpose(1, 2).reshape(B, T, H * D)
        for _ in range(5):   # repeat under load: a race shows as a run-to-run difference
            assert torch.equal(R.attention_x3(flat, H, scale, mode=mode), out), 'run-to-run difference'
        ref = (torch.softmax((q[:1].double() @ k[:1].double().transpose(-1, -2)) * scale, dim=-1) @ v[:1].double()).transpose(1, 2).reshape(1, T, H * D)
        e3, e32 = float((out[:1].double() - ref).abs().max()), float((o32[:1].double() - ref).abs().max())
        r3 = float((out[:1].double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
        r32 = float((o32[:1].double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
        t3 = bench(lambda: R.attention_x3(flat, H, scale, mode=mode))
        t32 = bench(lambda: F.scaled_dot_product_attention(q, k, v, scale=scale))
        fl = 4.0 * B * H * T * T * D
        print(f"B {B} T {T} H {H}: x3 {t3:.3f} ms = {fl / t3 / 1e9:.0f} TF-equiv ({6 * fl / t3 / 1e9:.0f} TF bf16 MFMA) | SDPA f32 {t32:.3f} ms = {fl / t32 / 1e9:.0f} TF"
              f" | max err: x3 {e3:.2e} f32 {e32:.2e} | rel rms: x3 {r3:.2e} f32 {r32:.2e}", flush=True)


if __name__ == "__main__":
    main()
