#!/usr/bin/env python3
"""GPU probe: DIBR-only frame time with float32 depth vs the reference's real input, an 8-bit depth-video frame (many identical
depth values -> same-address atomics in the pass-B kernels)."""
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
clip = [synth.synth_frame(i, H, W) for i in range(4)]
out = torch.empty((p.out_h, p.out_w, 3), dtype=torch.uint8, device="cuda")
for name, conv in (("f32 depth", lambda d: torch.from_numpy(d).cuda()), ("u8 gray depth", lambda d: torch.from_numpy(synth.depth_to_u8_bgr(d)[..., 0].copy()).cuda()),
                   ("u8 smooth (no noise)", lambda d: (torch.from_numpy(d).cuda() * 32).floor().div(32).mul(255).to(torch.uint8))):
    fs = [torch.from_numpy(f).cuda() for f, _ in clip]
    ds = [conv(d) for _, d in clip]
    r.reset_state(); r.new_clip()
    for i in range(12):
        r.render_frame(fs[i % 4], ds[i % 4], p, out=out)
    r.set_profiling(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 100
    for i in range(n):
        r.render_frame(fs[i % 4], ds[i % 4], p, out=out)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"{W}x{H} {name:22s}: {dt*1e3:.3f} ms/frame  select_eye {r.stage_ms('select_eye')*1e3:.0f} us  select_dc {r.stage_ms('select_dc')*1e3:.0f} us")
    r.set_profiling(False)
