"""Per-phase shader-cycle stamps of W1 (k_warp_fused) and E1 (k_finish_fused) at 4K, and the shader clock they actually ran at
(development build: bash tools/build_ab.sh stamps -DVD_PHASE_STAMPS; run on the GPU box with
VD3D_LIB_PATH=visiondepth3d_amd/ab/libvd3d_hip_stamps.so python tools/probe_phases.py).  Thread 0 of every 67th workgroup records
s_memtime at the barriers between phases; s_memrealtime (100 MHz) at the first and last stamp gives cycles / wall time = clock."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from visiondepth3d_amd import synth, _lib
from visiondepth3d_amd.params import render_kwargs_to_params
from visiondepth3d_amd.render_3d import Renderer

KW = dict(output_format="Half-SBS", fg_shift=10.0, mg_shift=-2.5, bg_shift=-5.0, sharpness_factor=0.15, dof_strength=2.0,
          feather_strength=10.0, blur_ksize=9, use_subject_tracking=True, use_floating_window=True)
H, W = 2160, 3840
if len(sys.argv) > 1 and sys.argv[1].startswith("gui"):   # round 6: the GUI's own defaults (no feathering: k_warp_fused<.., FEATHER = false>); "gui1080" at 1920x1080
    KW = dict(output_format="Full-SBS", fg_shift=4.5, mg_shift=-1.5, bg_shift=-6.0, sharpness_factor=0.2, dof_strength=2.0, feather_strength=0.0,
              blur_ksize=1, use_subject_tracking=True, use_floating_window=True, zero_parallax_strength=0.01)
    if sys.argv[1] == "gui1080":
        H, W = 1080, 1920
R = Renderer(0)
p = render_kwargs_to_params(W, H, output_height=H, **KW)
frames, depths = synth.synth_clip(4, H, W)
fr = [torch.from_numpy(f).cuda() for f in frames]
dp = [torch.from_numpy(d).cuda() for d in depths]
for i in range(8):
    R.render_frame(fr[i % 4], dp[i % 4], p)
torch.cuda.synchronize()
L = _lib.lib()
for name, fn, labels in (("W1 k_warp_fused", "vd3d_debug_stamps_w1", ["tables", "A warped depth", "B gradient mask", "C window sums", "D0 + column taps", "D1 Hh rows", "D2 sample + blend"]),
                         ("E1 k_finish_fused", "vd3d_debug_stamps_e1", ["tile load", "halo windows + blur weight", "dense levels", "blend + grade", "sharpen + fit + mux"])):
    buf = (C.c_ulonglong * 1024)()
    f = getattr(L, fn); f.argtypes = [C.c_void_p]
    assert f(buf) == 0
    t = np.array(buf, dtype=np.uint64).reshape(64, 16).astype(np.int64)
    n = len(labels)
    ok = [r for r in t if r[0] and r[n]]
    d = np.array([[r[k + 1] - r[k] for k in range(n)] for r in ok], dtype=np.float64)
    clk = np.array([(r[n] - r[0]) / ((r[15] - r[14]) / 100.0) for r in ok])
    tot = d.sum(axis=1)
    print(f"{name}: {len(ok)} sampled workgroups, {tot.mean():.0f} cycles each (min {tot.min():.0f}, max {tot.max():.0f}), shader clock {clk.mean():.0f} MHz (min {clk.min():.0f}, max {clk.max():.0f})")
    for k, lb in enumerate(labels):
        print(f"  {lb:28s} {d[:, k].mean():9.0f} cycles  {100 * d[:, k].mean() / tot.mean():5.1f} %   (min {d[:, k].min():.0f}, max {d[:, k].max():.0f})")

# residency: which workgroups overlapped in time on the same CU (HW_ID bits, gfx9 layout: wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13)
for name, fn in (("W1", "vd3d_debug_occ_w1"), ("E1", "vd3d_debug_occ_e1")):
    buf = (C.c_ulonglong * (16384 * 4))()
    f = getattr(L, fn); f.argtypes = [C.c_void_p]
    assert f(buf) == 0
    o = np.array(buf, dtype=np.uint64).reshape(16384, 4).astype(np.int64)
    o = o[(o[:, 2] > 0) & (o[:, 3] > 0)]
    hw, xcc = o[:, 0], o[:, 1] & 0xF
    cu = (xcc << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15)
    t0 = o[:, 2].min()
    print(f"{name} residency: {len(o)} workgroups on {len(np.unique(cu))} distinct CUs, launch span {(o[:, 3].max() - t0) / 100.0:.1f} us, mean workgroup life {np.mean(o[:, 3] - o[:, 2]) / 100.0:.2f} us")
    conc = []
    for c in np.unique(cu):
        sel = o[cu == c]
        ev = sorted([(a, 1) for a in sel[:, 2]] + [(b, -1) for b in sel[:, 3]])
        cur = mx = 0; area = 0; last = ev[0][0]
        for tm, dlt in ev:
            area += cur * (tm - last); last = tm
            cur += dlt; mx = max(mx, cur)
        conc.append((mx, area / max(1, ev[-1][0] - ev[0][0]), len(sel)))
    conc = np.array(conc, dtype=np.float64)
    print(f"  per CU: max concurrent workgroups {conc[:, 0].min():.0f} .. {conc[:, 0].max():.0f} (mean {conc[:, 0].mean():.2f}), time-averaged concurrency {conc[:, 1].mean():.2f}, workgroups per CU {conc[:, 2].min():.0f} .. {conc[:, 2].max():.0f}")
