"""Development probe: 4K (or any size) DIBR-only step throughput of the sharded / batched path for several launch shapes in ONE process
(the synthetic clip is built once).  python tools/probe_step.py [--size 2160x3840] [--B 16] [--steps 8] cfg ...   cfg = pix_streams:group:div[:w1_tile_height]
Prints pairs/s per configuration and the per-frame stage times (HIP events; profiling on = a second, untimed pass)."""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from visiondepth3d_amd import _lib, synth  # noqa: E402
from visiondepth3d_amd.params import render_kwargs_to_params  # noqa: E402
from visiondepth3d_amd.render_3d import Renderer  # noqa: E402
from visiondepth3d_amd.sharded import ChunkSharder, HipChunkBackend  # noqa: E402

KW = dict(output_format="Half-SBS", fg_shift=10.0, mg_shift=-2.5, bg_shift=-5.0, sharpness_factor=0.15, dof_strength=2.0,
          feather_strength=10.0, blur_ksize=9, use_subject_tracking=True, use_floating_window=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="2160x3840")
    ap.add_argument("--B", type=int, default=16)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--clip", type=int, default=16)
    ap.add_argument("--check", action="store_true", help="first: batched step == sequential render_frame, bit for bit")
    ap.add_argument("--dof", type=float, default=2.0)
    ap.add_argument("--fmt", default="Half-SBS")
    ap.add_argument("cfgs", nargs="*", default=["1:16:2"])
    a = ap.parse_args()
    sh, sw = [int(v) for v in a.size.split("x")]
    kw = dict(KW, dof_strength=a.dof, output_format=a.fmt)
    p = render_kwargs_to_params(sw, sh, output_height=sh, **kw)
    t0 = time.perf_counter()
    fr, dp = synth.synth_clip(a.clip, sh, sw)
    frames = torch.stack([torch.from_numpy(f) for f in fr]).cuda()
    depths = torch.stack([torch.from_numpy(d) for d in dp]).cuda()
    print(f"clip {a.clip} x {sw}x{sh} built in {time.perf_counter() - t0:.1f} s", flush=True)
    B = a.B
    outs = torch.empty((B, p.out_h, p.out_w, 3), dtype=torch.uint8, device="cuda")
    L = _lib.lib()
    if a.check:
        r = Renderer(0)
        r.new_clip()
        seq = [r.render_frame(frames[j % a.clip], depths[j % a.clip], p).clone() for j in range(2 * B)]
        r.reset_state(); r.new_clip()
        be = HipChunkBackend(r, p)
        s0 = ChunkSharder(be, 0, 1, B)
        s1 = ChunkSharder(be, 0, 1, B, slot_base=B, twin_of=s0)
        r.set_pixel_overlap(2)
        got = []
        for i, s in enumerate((s0, s1)):
            idx = [(i * B + j) % a.clip for j in range(B)]
            o = s.render_step([frames[k] for k in idx], [depths[k] for k in idx], first_step=(i == 0))
            got += o
        r.sync()
        bad = [j for j in range(2 * B) if not torch.equal(got[j], seq[j])]
        print("CHECK batched+2 pixel streams vs sequential:", "OK" if not bad else f"MISMATCH frames {bad}", flush=True)
        r.set_pixel_overlap(0)
        r.close()
    for cfg in a.cfgs:
        ps, grp, div, *rest = [int(v) for v in cfg.split(":")]
        L.vd3d_debug_tune(0, div)
        L.vd3d_debug_tune(1, grp)
        L.vd3d_debug_tune(2, rest[0] if rest else 32)   # W1 tile height (precomputed-mask variants)
        r = Renderer(0)
        r.new_clip()
        be = HipChunkBackend(r, p)
        sets = [ChunkSharder(be, 0, 1, B)]
        if ps > 0:
            sets.append(ChunkSharder(be, 0, 1, B, slot_base=B, twin_of=sets[0]))
            r.set_pixel_overlap(ps)

        def step(i):
            s = sets[i % len(sets)]
            idx = [(i * B + j) % a.clip for j in range(B)]
            fl = frames[idx[0]:idx[0] + B] if idx == list(range(idx[0], idx[0] + B)) else frames[idx]
            dl = depths[idx[0]:idx[0] + B] if idx == list(range(idx[0], idx[0] + B)) else depths[idx]
            s.p1(fl, dl, first_step=(i == 0)); s.r1(s.gather(s.q_local)); s.p3(); s.r2(s.gather(s.m_local)); s.pixels(outs)
        for i in range(3):
            step(i)
        r.sync(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(3, 3 + a.steps):
            step(i)
        r.sync(); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        r.set_profiling(True)
        for i in range(3 + a.steps, 5 + a.steps):
            step(i)
        st = {k: round(r.stage_ms(k) * 1e3, 1) for k in ("p1_own", "p3_own", "replay", "shift", "w1", "finish")}
        r.set_profiling(False)
        print(f"cfg pix_streams={ps} group={grp} div={div} w1th={rest[0] if rest else 32}: {a.steps * B / dt:8.1f} pairs/s  {dt / (a.steps * B) * 1e6:7.1f} us/frame   stage us per call {st}", flush=True)
        if ps > 0:
            r.set_pixel_overlap(0)
        r.close()


if __name__ == "__main__":
    main()
