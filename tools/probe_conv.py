"""Times vd3d_conv3x3_c64_f16 against MIOpen's fp16 convolution + PReLU on a 960x540x64 activation (the body layer of the up-scale
network at configs[4]'s size).  usage (GPU box): python tools/probe_conv.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from visiondepth3d_amd.render_3d import Renderer
from visiondepth3d_amd.upscale import conv_weight_fragments, Upscaler

R = Renderer(0)
H, W = 540, 960
x = (torch.randn(1, 64, H, W, device="cuda") * 0.5).half().contiguous(memory_format=torch.channels_last)
w = (torch.randn(64, 64, 3, 3, device="cuda") * 0.05).half()
b = torch.randn(64, device="cuda") * 0.1
sl = torch.rand(64, device="cuda") * 0.3
wf = conv_weight_fragments(w).cuda()
y = torch.empty_like(x, memory_format=torch.channels_last)
flops = 2.0 * H * W * 64 * 64 * 9


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return a.elapsed_time(e) / n


t_hip = timeit(lambda: R.conv3x3_c64(x, wf, b, sl, out=y))
wc = w.contiguous(memory_format=torch.channels_last)
bh, slh = b.half(), sl.half()
t_lib = timeit(lambda: F.prelu(F.conv2d(x, wc, bh, padding=1), slh))
print(f"conv3x3 64->64 {W}x{H}: hip {t_hip*1e3:.1f} us ({flops/t_hip/1e9:.1f} TFLOP/s, {flops/t_hip/1e9/2500*100:.1f} % of fp16 peak) | "
      f"MIOpen conv + prelu {t_lib*1e3:.1f} us ({flops/t_lib/1e9:.1f} TFLOP/s)")
for hb in (True, False):
    up = Upscaler(R, "RealESR_Gx4_fp16", hip_body=hb)
    f = torch.randint(0, 256, (H, W, 3), dtype=torch.uint8, device="cuda")
    t = timeit(lambda: up._infer(f), 5)
    print(f"RealESR_Gx4 forward {W}x{H} hip_body={hb}: {t:.3f} ms")
