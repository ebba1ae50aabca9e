"""CPU-only cross-check of the oracle against the LIVE reference (``/root/reference`` imported through tests/golden/ref_loader.py
with the cv2 / torchvision / tkinter stand-ins of ref_stubs.py) over a seeded random sweep of ``pixel_shift_cuda`` parameters
and sizes.  Complements the committed goldens: the fixtures pin fixed cases everywhere, this sweep widens the pinned region
wherever the reference tree is present (the development container).  Skipped when it is not (the GPU box: by rule nothing there
may read /root/reference)."""
import os
import sys

import numpy as np
import pytest

from conftest import u8_diff_stats
from visiondepth3d_amd import synth
from visiondepth3d_amd._abi import ShiftParams, State

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import ref_loader  # noqa: E402

pytestmark = pytest.mark.skipif(not ref_loader.available(), reason="reference tree not present")


@pytest.fixture(scope="module")
def ref():
    import torch
    torch.set_num_threads(4)
    return ref_loader.load()


@pytest.mark.parametrize("seed", range(24))
def test_pixel_shift_random_parameters(ref, oracle, seed):
    import torch
    rng = np.random.default_rng(5000 + seed)
    ih, iw = int(rng.integers(24, 80)), int(rng.integers(32, 130))
    if rng.integers(0, 3) == 0:
        H, W = ih, iw
    else:
        H, W = int(rng.integers(24, 120)), int(rng.integers(32, 200))
    kw = dict(blur_ksize=int(rng.integers(0, 6)) * 2 + 1, feather_strength=float(rng.uniform(0, 20)),   # odd k: even k changes the plane size in torch
              use_subject_tracking=bool(rng.integers(0, 2)), enable_floating_window=bool(rng.integers(0, 2)),
              max_pixel_shift_percent=float(rng.uniform(0.005, 0.06)), zero_parallax_strength=float(rng.uniform(0, 0.03)),
              enable_edge_masking=bool(rng.integers(0, 3) > 0), enable_feathering=bool(rng.integers(0, 3) > 0),
              convergence_strength=float([0.0, 3.0, -2.0][int(rng.integers(0, 3))]), enable_dynamic_convergence=bool(rng.integers(0, 2)),
              depth_pop_gamma=float(rng.uniform(0.6, 1.3)), depth_pop_mid=float(rng.uniform(0.35, 0.65)),
              parallax_balance=float(rng.uniform(0.5, 1.0)))
    fg, mg, bg = float(rng.uniform(0, 30)), float(rng.uniform(-10, 5)), float(rng.uniform(-25, 0))
    bgr, d = synth.synth_frame(seed, ih, iw)
    ft = oracle.frame_to_tensor(bgr)
    ref_loader.reset_state(ref)
    with torch.no_grad():
        rl, rr, rs = ref.pixel_shift_cuda(torch.from_numpy(ft), torch.from_numpy(d[None].copy()), W, H, fg, mg, bg, return_shift_map=True, **kw)
    st = State()
    o = oracle.pixel_shift(ft, d[None], W, H, ShiftParams.defaults(fg, mg, bg, **kw), st, want_shift=True)
    assert st.fw_prev_offset == ref.floating_window_tracker.prev_offset, (seed, kw)
    # shift map: the reference's pow / exp / sigmoid are SLEEF 1-ULP kernels, the oracle's are correctly rounded; one ULP of a
    # layer weight is amplified by (1.2 fg + |mg| + 1.1 |bg|) * balance / (W/2).  5e-7 in normalised units is < 1e-4 pixel here.
    assert np.max(np.abs(o["shift"] - rs.numpy())) < 5e-7, (seed, kw)
    for got, exp, eye in ((o["left"], rl, "L"), (o["right"], rr, "R")):
        mx, frac, _ = u8_diff_stats(got, np.asarray(exp))
        assert mx <= 1 and frac < 3e-3, (seed, eye, mx, frac, kw)      # <= 1 LSB (pow / exp 1-ULP differences on truncation cliffs)


@pytest.mark.parametrize("seed", range(8))
def test_render_loop_random_configurations(ref, oracle, seed):
    """The real ``render_sbs_3d`` loop of the live reference (fake VideoCapture / VideoWriter of ref_stubs) vs the oracle on
    random configurations; bars of the committed B2 goldens (tests/test_oracle_vs_golden.py)."""
    import make_golden as mg   # importable: generation only runs under __main__
    from visiondepth3d_amd.params import render_kwargs_to_params
    rng = np.random.default_rng(7000 + seed)
    fmt = ["Half-SBS", "Full-SBS", "Passive Interlaced", "Red-Cyan Anaglyph"][int(rng.integers(0, 4))]
    sh = int(rng.integers(40, 100)) // 2 * 2
    sw = int(round(sh * [16 / 9, 16 / 9, 4 / 3, 2.0][int(rng.integers(0, 4))])) // 2 * 2
    kw = dict(output_format=fmt, output_height=sh, fg_shift=float(rng.uniform(2, 20)), mg_shift=float(rng.uniform(-6, 2)),
              bg_shift=float(rng.uniform(-15, 0)), sharpness_factor=float(rng.uniform(0.0, 0.4)),
              dof_strength=float([0.0, 1.0, 2.0][int(rng.integers(0, 3))]), feather_strength=float(rng.uniform(0, 15)),
              blur_ksize=int(rng.integers(0, 5)) * 2 + 1, use_subject_tracking=bool(rng.integers(0, 2)),
              use_floating_window=bool(rng.integers(0, 2)), ipd_factor=float([1.0, 1.0, 1.2][int(rng.integers(0, 3))]),
              color_saturation=float(rng.uniform(0.9, 1.3)), color_contrast=float(rng.uniform(0.95, 1.1)),
              color_brightness=float(rng.uniform(-0.03, 0.03)))
    if fmt == "Full-SBS":
        kw.update(preserve_original_aspect=True, original_video_width=sw, original_video_height=sh)
    n = 4
    name = f"_live_{seed}"
    mg.LOOP_CASES[name] = (sh, sw, n, kw)
    try:
        written = np.stack(mg.run_loop(name))
    finally:
        del mg.LOOP_CASES[name]
    frames, depths = synth.synth_clip(n, sh, sw)
    ro = oracle.RenderOracle(render_kwargs_to_params(sw, sh, **kw))
    ro.new_clip()
    got = np.stack([ro.render(f, synth.depth_to_u8_bgr(d), 1) for f, d in list(zip(frames, depths))[1:]])
    assert got.shape == written.shape, (got.shape, written.shape, kw)
    mx, frac, frac_gt1 = u8_diff_stats(got, written)
    assert mx <= 8 and frac_gt1 < 5e-3 and frac < 1.5e-2, (seed, mx, frac, frac_gt1, kw)
