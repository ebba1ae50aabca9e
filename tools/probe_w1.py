#!/usr/bin/env python3
"""GPU probe: time the fused warp kernel (stage "w1") alone for feather on/off x resize/identity (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visiondepth3d_amd import synth
from visiondepth3d_amd._abi import ShiftParams
from visiondepth3d_amd.render_3d import Renderer

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (2160, 3840)
r = Renderer(0)
f, d = synth.synth_frame(0, H, W)
ft = (torch.from_numpy(f).cuda().flip(-1).permute(2, 0, 1).float() / 255).contiguous()
dt = torch.from_numpy(d).cuda()[None]
fe = torch.nn.functional.interpolate(ft[None], size=(H // 2, W // 2), mode="bilinear")[0].contiguous()
de = torch.nn.functional.interpolate(dt[None], size=(H // 2, W // 2), mode="bilinear")[0].contiguous()
for name, (a, b) in {"resize(2x)": (fe, de), "identity": (ft, dt)}.items():
    for feather, k in ((1, 9), (1, 1), (0, 9)):
        p = ShiftParams.defaults(10, -2.5, -5, enable_feathering=feather, blur_ksize=k)
        for _ in range(3):
            r.pixel_shift(a, b, W, H, p)
        r.set_profiling(True)
        for _ in range(10):
            r.pixel_shift(a, b, W, H, p)
        print(f"{W}x{H} {name:10s} feather={feather} k={k}: w1 {r.stage_ms('w1')*1e3:7.1f} us  shift {r.stage_ms('shift')*1e3:6.1f} us  "
              f"pixel_shift total {r.stage_ms('pixel_shift')*1e3:7.1f} us", flush=True)
        r.set_profiling(False)
