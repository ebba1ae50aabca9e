"""Experiment: the conv kernel as two half-image launches on two streams (workgroups of the two launches share CUs out of phase) vs one
launch.  Timing only (the halves are treated as independent images).  usage (GPU box): python tools/probe_conv2.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visiondepth3d_amd.render_3d import Renderer
from visiondepth3d_amd.upscale import conv_weight_fragments

H, W = 544, 960
x = (torch.randn(1, 64, H, W, device="cuda") * 0.5).half().contiguous(memory_format=torch.channels_last)
w = (torch.randn(64, 64, 3, 3, device="cuda") * 0.05).half()
b = torch.randn(64, device="cuda") * 0.1
sl = torch.rand(64, device="cuda") * 0.3
wf = conv_weight_fragments(w).cuda()
y = torch.empty_like(x, memory_format=torch.channels_last)
R0 = Renderer(0)
Ra, Rb = Renderer(0, private_stream=True, auto_order=False), Renderer(0, private_stream=True, auto_order=False)
xs = x.permute(0, 2, 3, 1)   # [1,H,W,64] view of the NHWC memory
ys = y.permute(0, 2, 3, 1)
halves = []
for k in range(2):
    xh = xs[:, k * H // 2:(k + 1) * H // 2].permute(0, 3, 1, 2)
    yh = ys[:, k * H // 2:(k + 1) * H // 2].permute(0, 3, 1, 2)
    assert xh.is_contiguous(memory_format=torch.channels_last)
    halves.append((xh, yh))


def t_one(n=20):
    for _ in range(3):
        R0.conv3x3_c64(x, wf, b, sl, out=y)
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        R0.conv3x3_c64(x, wf, b, sl, out=y)
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) / n


def t_two(n=20):
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for i in range(n + 3):
        if i == 3:
            torch.cuda.synchronize()
            a.record(Ra.stream); 
        Ra.conv3x3_c64(halves[0][0], wf, b, sl, out=halves[0][1])
        Rb.conv3x3_c64(halves[1][0], wf, b, sl, out=halves[1][1])
    Ra.sync(); 
    e.record(Rb.stream); Rb.sync(); torch.cuda.synchronize()
    return a.elapsed_time(e) / n


print(f"one launch {t_one()*1e3:.1f} us; two half launches on two streams {t_two()*1e3:.1f} us")
