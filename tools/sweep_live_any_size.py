"""Offline sweeps behind DESIGN.md section 2 "N-thread ATen mode" (development container only: imports /root/reference through tests/golden/ref_loader.py).

    python tools/sweep_live_any_size.py loop  A B [threads] [odd]   # render_sbs_3d at random frame sizes / aspects, seeds A .. B-1, oracle in the N-thread mode
    python tools/sweep_live_any_size.py wide  A B [threads]         # the same with other output heights (fits), skip_blank_frames, black-bar auto crop on top
    python tools/sweep_live_any_size.py shift A B [threads]         # pixel_shift_cuda at odd sizes (the body of test_pixel_shift_random_parameters), exact comparison

Prints one line per configuration that differs and the count at the end.  The runs of round 5 and their results: profiles/r05_parity_sweeps.md."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden"), ROOT]
import make_golden as mg  # noqa: E402
import ref_loader  # noqa: E402
from oracle import oracle  # noqa: E402
from visiondepth3d_amd import synth  # noqa: E402
from visiondepth3d_amd._abi import ShiftParams, State  # noqa: E402
from visiondepth3d_amd.params import render_kwargs_to_params  # noqa: E402


def loop_case(seed, odd):
    sh, sw, kw = mg.any_size_case(seed)
    if odd:   # the odd-size branch for every seed (any_size_case alternates by seed parity); VD3D_SWEEP_SCALE=k: k times larger frames (planes above 32 768 elements:
        # several chunks per plane, each with its own tail)
        rng = np.random.default_rng(97000 + seed)
        k = int(os.environ.get("VD3D_SWEEP_SCALE", "1"))
        sh, sw = int(rng.integers(40 * k, 150 * k)), int(rng.integers(60 * k, 260 * k))
        kw["output_height"] = sh
        if "original_video_width" in kw:
            kw.update(original_video_width=sw, original_video_height=sh)
    return sh, sw, kw


def sweep_loop(a, b, threads, odd):
    bad = 0
    for seed in range(a, b):
        sh, sw, kw = loop_case(seed, odd)
        name = f"_sweep_any_{seed}"
        mg.LOOP_CASES[name] = (sh, sw, 4, kw)
        try:
            written = np.stack(mg.run_loop(name))
        finally:
            del mg.LOOP_CASES[name]
        frames, depths = synth.synth_clip(4, sh, sw)
        p = render_kwargs_to_params(sw, sh, aten_sum_threads=threads, **kw)
        ro = oracle.RenderOracle(p)
        ro.new_clip()
        got = np.stack([ro.render(f, synth.depth_to_u8_bgr(d), 1) for f, d in list(zip(frames, depths))[1:]])
        d = -1 if got.shape != written.shape else int((got != written).sum())
        if d:
            bad += 1
            print("seed", seed, (sh, sw), kw["output_format"], "eye", (p.eye_h, p.eye_w), "warp", (p.warp_h, p.warp_w), "differing samples", d)
    return bad


def sweep_wide(a, b, threads):
    """render_sbs_3d at random sizes with the loop-level variants on top: output heights that differ from the source (INTER_AREA / INTER_LINEAR fits), skip_blank_frames
    with a random blank list, auto_crop_black_bars on letterboxed clips.  Configurations whose fit the oracle does not restate are counted as skipped."""
    import contextlib
    import io
    import threading
    bad = skipped = 0
    for seed in range(a, b):
        sh, sw, kw = loop_case(seed, True)
        rng = np.random.default_rng(555000 + seed)
        variant = int(rng.integers(0, 4))
        n = 6
        blank = []
        frames = dbgr = None
        if variant in (0, 3):
            kw["output_height"] = max(24, int(sh * float(rng.uniform(0.5, 2.0))))
        if variant in (1, 3):
            kw["skip_blank_frames"] = True
            blank = sorted(set(int(v) for v in rng.integers(0, n - 1, size=int(rng.integers(1, 4)))))
        if variant == 2 and sh >= 60:
            kw["auto_crop_black_bars"] = True
            frames, dbgr = synth.letterbox_clip(n, sh, sw, int(rng.integers(2, sh // 6)), int(rng.integers(2, sh // 6)))
        if frames is None:
            frames, depths = synth.synth_clip(n, sh, sw)
            dbgr = [synth.depth_to_u8_bgr(d) for d in depths]
        mg.ref_stubs._Clip.clips["in.mp4"] = frames
        mg.ref_stubs._Clip.clips["depth.mp4"] = dbgr
        mg.rl.reset_state()
        args = dict(input_path="in.mp4", depth_path="depth.mp4", output_path="out.avi", selected_codec="XVID", fps=24.0, output_width=sw,
                    selected_aspect_ratio=mg._Aspect("Default (16:9)"), aspect_ratios=mg.r.aspect_ratios, suspend_flag=threading.Event(), cancel_flag=threading.Event())
        args.update(kw)
        saved = mg.r.detect_black_white_frames
        mg.r.detect_black_white_frames = lambda *a_, **k_: list(blank)
        try:
            with contextlib.redirect_stdout(io.StringIO()) as so:
                mg.r.render_sbs_3d(**args)
        finally:
            mg.r.detect_black_white_frames = saved
        if "crashed" in so.getvalue():
            print("seed", seed, "reference crashed:", so.getvalue()[-200:].replace("\n", " | ")); skipped += 1; continue
        written = np.stack(mg.ref_stubs._Clip.written["out.avi"])
        try:
            p = render_kwargs_to_params(sw, sh, aten_sum_threads=threads, **kw)
            ro = oracle.RenderOracle(p)
            ro.new_clip()
            got = np.stack([ro.render(f, d, 1, blank=(i in blank)) for i, (f, d) in enumerate(list(zip(frames, dbgr))[1:])])
        except NotImplementedError as e:
            skipped += 1
            continue
        d = -1 if got.shape != written.shape else int((got != written).sum())
        if d:
            bad += 1
            print("seed", seed, "variant", variant, (sh, sw), kw["output_format"], "out_h", kw["output_height"], "blank", blank, "eye", (p.eye_h, p.eye_w), "warp", (p.warp_h, p.warp_w),
                  "differing samples", d, "max", int(np.abs(got.astype(int) - written).max()) if d > 0 else None, "frames", [int((got[i] != written[i]).sum()) for i in range(len(got))] if d > 0 else None)
    print("skipped (fits the oracle does not restate / reference errors):", skipped)
    return bad


def sweep_shift(a, b, threads, ref):
    bad = 0
    for seed in range(a, b):
        rng = np.random.default_rng(5000 + seed)
        k = int(os.environ.get("VD3D_SWEEP_SCALE", "1"))   # k times larger planes (several chunks per plane)
        ih, iw = int(rng.integers(24 * k, 80 * k)), int(rng.integers(32 * k, 130 * k))
        H, W = (ih, iw) if rng.integers(0, 3) == 0 else (int(rng.integers(24 * k, 120 * k)), int(rng.integers(32 * k, 200 * k)))
        kw = dict(blur_ksize=int(rng.integers(0, 6)) * 2 + 1, feather_strength=float(rng.uniform(0, 20)),
                  use_subject_tracking=bool(rng.integers(0, 2)), enable_floating_window=bool(rng.integers(0, 2)),
                  max_pixel_shift_percent=float(rng.uniform(0.005, 0.06)), zero_parallax_strength=float(rng.uniform(0, 0.03)),
                  enable_edge_masking=bool(rng.integers(0, 3) > 0), enable_feathering=bool(rng.integers(0, 3) > 0),
                  convergence_strength=float([0.0, 3.0, -2.0][int(rng.integers(0, 3))]), enable_dynamic_convergence=bool(rng.integers(0, 2)),
                  depth_pop_gamma=float(rng.uniform(0.6, 1.3)), depth_pop_mid=float(rng.uniform(0.35, 0.65)),
                  parallax_balance=float(rng.uniform(0.5, 1.0)))
        fg, mgs, bg = float(rng.uniform(0, 30)), float(rng.uniform(-10, 5)), float(rng.uniform(-25, 0))
        bgr, d = synth.synth_frame(seed, ih, iw)
        ft = oracle.frame_to_tensor(bgr)
        ref_loader.reset_state(ref)
        with torch.no_grad():
            rl, rr, rs = ref.pixel_shift_cuda(torch.from_numpy(ft), torch.from_numpy(d[None].copy()), W, H, fg, mgs, bg, return_shift_map=True, **kw)
        o = oracle.pixel_shift(ft, d[None], W, H, ShiftParams.defaults(fg, mgs, bg, aten_threads=threads, **kw), State(), want_shift=True)
        ds = int((o["shift"].view(np.uint32) != rs.numpy().view(np.uint32)).sum())
        dl, dr = int((o["left"] != np.asarray(rl)).sum()), int((o["right"] != np.asarray(rr)).sum())
        if ds or dl or dr:
            bad += 1
            print("seed", seed, (ih, iw, H, W), "shift", ds, "left", dl, "right", dr)
    return bad


if __name__ == "__main__":
    mode, a, b = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    threads = int(sys.argv[4]) if len(sys.argv) > 4 else 4
    ref = ref_loader.load()
    torch.set_num_threads(threads)
    assert torch.get_num_threads() == threads
    n_bad = sweep_loop(a, b, threads, len(sys.argv) > 5) if mode == "loop" else sweep_wide(a, b, threads) if mode == "wide" else sweep_shift(a, b, threads, ref)
    print(f"{mode} seeds {a}..{b - 1} at {threads} torch threads: {b - a - n_bad} of {b - a} exact")
