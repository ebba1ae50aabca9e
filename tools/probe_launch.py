#!/usr/bin/env python3
"""GPU probe: host enqueue time vs GPU time of Renderer.render_frame (is the 1080p DIBR-only path launch-bound?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visiondepth3d_amd import synth
from visiondepth3d_amd.params import render_kwargs_to_params
from visiondepth3d_amd.render_3d import Renderer
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1080, 1920)
p = render_kwargs_to_params(W, H, output_height=H, output_format="Half-SBS", fg_shift=10.0, mg_shift=-2.5, bg_shift=-5.0, sharpness_factor=0.15,
                            dof_strength=2.0, feather_strength=10.0, blur_ksize=9, use_subject_tracking=True, use_floating_window=True)
r = Renderer(0)
f, d = synth.synth_frame(0, H, W)
ft, dt = torch.from_numpy(f).cuda(), torch.from_numpy(d).cuda()
out = torch.empty((p.out_h, p.out_w, 3), dtype=torch.uint8, device="cuda")
for _ in range(20):
    r.render_frame(ft, dt, p, out=out)
torch.cuda.synchronize()
n = 300
t0 = time.perf_counter()
for _ in range(n):
    r.render_frame(ft, dt, p, out=out)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"{W}x{H}: host enqueue {1e3*(t1-t0)/n:.3f} ms/frame, total {1e3*(t2-t0)/n:.3f} ms/frame ({n/(t2-t0):.0f} pairs/s)")
