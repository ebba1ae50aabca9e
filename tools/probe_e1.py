#!/usr/bin/env python3
"""GPU probe: time the fused finishing kernel (stage "finish") alone (development aid).
usage: probe_e1.py [H W] [format]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visiondepth3d_amd import synth
from visiondepth3d_amd.params import render_kwargs_to_params
from visiondepth3d_amd.render_3d import Renderer

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2160, 3840)
fmt = sys.argv[3] if len(sys.argv) > 3 else "Half-SBS"
r = Renderer(0)
p = render_kwargs_to_params(W, H, output_height=H, fg_shift=8.0, mg_shift=-3.0, bg_shift=-6.0, sharpness_factor=0.2, output_format=fmt,
                            dof_strength=2.0)
f, d = synth.synth_frame(0, H, W)
wh, ww, eh, ew = p.warp_h, p.warp_w, p.eye_h, p.eye_w
g = torch.Generator().manual_seed(1)
L = torch.from_numpy(f).cuda()
L = torch.nn.functional.interpolate(L.permute(2, 0, 1)[None].float(), size=(wh, ww), mode="bilinear")[0].permute(1, 2, 0).to(torch.uint8).contiguous()
R = L.flip(1).contiguous()
dn = torch.nn.functional.interpolate(torch.from_numpy(d).cuda()[None, None], size=(eh, ew), mode="bilinear")[0, 0].contiguous()
for _ in range(3):
    r.finish_frame(L, R, dn, p, 0.5)
r.set_profiling(True)
for _ in range(20):
    r.finish_frame(L, R, dn, p, 0.5)
print(f"{fmt} warp {ww}x{wh} eye {ew}x{eh}: finish {r.stage_ms('finish')*1e3:7.1f} us", flush=True)
