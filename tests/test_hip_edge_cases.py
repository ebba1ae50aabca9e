"""GPU edge cases the reference's semantics make possible: odd / tiny sizes, even and large feather kernels, huge shift
bounds (fused kernels fall back to the one-stage kernels), every fit path (identity, 2:1, 2x2 SIMD rounding, padded
canvas), DOF radii beyond the fused fast path, collapse / few-valid planes, loud failures for what is not built."""
import numpy as np
import pytest

from conftest import u8_diff_stats
from visiondepth3d_amd import synth
from visiondepth3d_amd._abi import ShiftParams, State
from visiondepth3d_amd._lib import Vd3dError
from visiondepth3d_amd.params import render_kwargs_to_params

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def R():
    from visiondepth3d_amd.render_3d import Renderer
    r = Renderer(0)
    yield r
    r.close()


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _shift_eq(R, oracle, ih, iw, H, W, fg=10.0, mg=-2.5, bg=-5.0, idx=1, **kw):
    bgr, d = synth.synth_frame(idx, ih, iw)
    ft = oracle.frame_to_tensor(bgr)
    p = ShiftParams.defaults(fg, mg, bg, **kw)
    st = State()
    o = oracle.pixel_shift(ft, d[None], W, H, p, st, want_shift=True)
    R.reset_state()
    L, Rr, S = R.pixel_shift(T(ft), T(d[None]), W, H, p, want_shift=True)
    assert np.array_equal(S.cpu().numpy(), o["shift"]), (ih, iw, H, W, kw)
    assert np.array_equal(L.cpu().numpy(), o["left"]), (ih, iw, H, W, kw, u8_diff_stats(L.cpu().numpy(), o["left"]))
    assert np.array_equal(Rr.cpu().numpy(), o["right"]), (ih, iw, H, W, kw)
    assert R.export_state().fw_prev_offset == st.fw_prev_offset


@pytest.mark.parametrize("sizes", [(37, 53, 37, 53), (20, 30, 61, 95), (64, 64, 65, 127), (33, 70, 99, 210), (90, 160, 45, 80),
                                   (8, 8, 8, 8), (50, 67, 67, 50)])
def test_pixel_shift_odd_sizes_and_ratios(R, oracle, sizes):
    """identity, non-integer up- and down-scaling, transposed aspect, tiles that do not divide the image."""
    _shift_eq(R, oracle, *sizes)


@pytest.mark.parametrize("kw", [dict(blur_ksize=2), dict(blur_ksize=4, feather_strength=3.0), dict(blur_ksize=15, feather_strength=20.0),
                                dict(blur_ksize=33), dict(blur_ksize=35), dict(blur_ksize=64, feather_strength=4.0), dict(blur_ksize=129),
                                dict(blur_ksize=1, feather_strength=0.0),
                                dict(max_pixel_shift_percent=0.3),                       # bound too large for the fused LDS tiles -> fallback
                                dict(max_pixel_shift_percent=0.12, blur_ksize=21),
                                dict(convergence_strength=5.0), dict(convergence_strength=-3.0, enable_dynamic_convergence=False),
                                dict(enable_feathering=False, enable_edge_masking=False, use_subject_tracking=False)])
def test_pixel_shift_parameter_extremes(R, oracle, kw):
    _shift_eq(R, oracle, 54, 96, 108, 192, **kw)
    _shift_eq(R, oracle, 72, 128, 72, 128, fg=30.0, mg=-8.0, bg=-20.0, **kw)


def test_blur_ksize_limits(R):
    f = torch.zeros(3, 16, 16).cuda()
    d = torch.zeros(1, 16, 16).cuda()
    with pytest.raises(Vd3dError):
        R.pixel_shift(f, d, 16, 16, ShiftParams.defaults(1, 1, 1, blur_ksize=0))      # avg_pool2d raises in the reference
    with pytest.raises(Vd3dError) as e:
        R.pixel_shift(f, d, 16, 16, ShiftParams.defaults(1, 1, 1, blur_ksize=131))    # valid in the reference, beyond the 160 KB LDS tile: loud
    assert e.value.code == -4


def _loop_eq(R, oracle, sh, sw, n, kw, depth_fmt="f32"):
    p = render_kwargs_to_params(sw, sh, **kw)
    frames, depths = synth.synth_clip(n, sh, sw)
    R.reset_state(); R.new_clip()
    ro = oracle.RenderOracle(p); ro.new_clip()
    for i, (f, d) in enumerate(zip(frames, depths)):
        dd = d if depth_fmt == "f32" else synth.depth_to_u8_bgr(d)[..., 0].copy()
        got = R.render_frame(T(f), T(dd), p).cpu().numpy()
        exp = ro.render(f, dd, 0 if depth_fmt == "f32" else 2)
        assert R.last_scalars().as_dict() == ro.last.as_dict(), (i, kw)
        assert np.array_equal(got, exp), (i, kw, u8_diff_stats(got, exp))
    return p


BASE = dict(fg_shift=10.0, mg_shift=-2.5, bg_shift=-5.0, sharpness_factor=0.15, dof_strength=2.0, feather_strength=10.0, blur_ksize=9,
            use_subject_tracking=True, use_floating_window=True)


def test_fit_paths(R, oracle):
    # Half-SBS 2:1 box (fused), gray uint8 depth
    _loop_eq(R, oracle, 72, 128, 3, dict(BASE, output_format="Half-SBS", output_height=72), depth_fmt="gray")
    # Full-SBS without preserve: eyes hard-wired to 1920x1080 (:1120-1123); warp at 3840x2160 would be the 2x2 SIMD path --
    # exercised here at a small size through preserve + an explicit 2x down-fit is not reachable, so cover the 2x2 rounding
    # with the finish_frame entry point instead (below).  Full-SBS preserve = identity fit:
    _loop_eq(R, oracle, 54, 96, 2, dict(BASE, output_format="Full-SBS", output_height=54, preserve_original_aspect=True,
                                        original_video_width=96, original_video_height=54))
    # interlaced / anaglyph (anaglyph runs the unfused finish kernels)
    _loop_eq(R, oracle, 54, 96, 2, dict(BASE, output_format="Passive Interlaced", output_height=54))
    _loop_eq(R, oracle, 54, 96, 2, dict(BASE, output_format="Red-Cyan Anaglyph", output_height=54))
    # 4:3 source centre-cropped to 16:9
    _loop_eq(R, oracle, 120, 160, 2, dict(BASE, output_format="Half-SBS", output_height=90))


@pytest.mark.parametrize("fit", [(64, 36, "2x2"), (128, 72, "id"), (64, 72, "2x1"), (32, 18, "4x4"), (128, 36, "1x2")])
def test_finish_frame_integer_fits(R, oracle, fit):
    """vd3d_finish_frame: every integer INTER_AREA ratio incl. OpenCV's 2x2 (a+b+c+d+2)>>2 path and a padded canvas."""
    fw, fh, _ = fit
    H, W = 72, 128
    L_, d = synth.synth_frame(2, H, W)
    R_ = synth.synth_frame(5, H, W)[0]
    p = render_kwargs_to_params(W, H, **dict(BASE, output_format="Full-SBS", output_height=H, preserve_original_aspect=True,
                                             original_video_width=W, original_video_height=H))
    p.fit_w, p.fit_h, p.out_w, p.out_h = fw, fh, 2 * fw, fh
    dn = synth.synth_frame(2, H // 2, W // 2)[1]
    exp = oracle.finish_frame(L_, R_, dn, p, 0.37, 5, 1)
    got = R.finish_frame(T(L_), T(R_), T(dn), p, 0.37, 5, 1).cpu().numpy()
    assert np.array_equal(got, exp), (fit, u8_diff_stats(got, exp))


def test_finish_frame_padded_canvas_and_unsupported_ratio(R, oracle):
    H, W = 54, 128   # wider than the 16:9 canvas -> letterboxed by pad_to_aspect_ratio (:101-131)
    L_ = synth.synth_frame(1, H, W)[0]
    R_ = synth.synth_frame(4, H, W)[0]
    dn = synth.synth_frame(1, H, W)[1]
    p = render_kwargs_to_params(W, H, **dict(BASE, output_format="Full-SBS", output_height=H, preserve_original_aspect=True,
                                             original_video_width=W, original_video_height=H))
    p.fit_w, p.fit_h, p.out_w, p.out_h = 128, 72, 256, 72
    exp = oracle.finish_frame(L_, R_, dn, p, 0.5, 0, 0)
    got = R.finish_frame(T(L_), T(R_), T(dn), p, 0.5, 0, 0).cpu().numpy()
    assert np.array_equal(got, exp)
    assert not got[:9].any() and not got[63:].any()           # black bars of the canvas
    p.fit_w, p.fit_h, p.out_w, p.out_h = 100, 72, 200, 72      # fractional INTER_AREA (1.28): generic area-table path
    got = R.finish_frame(T(L_), T(R_), T(dn), p, 0.5, 0, 0).cpu().numpy()
    assert np.array_equal(got, oracle.finish_frame(L_, R_, dn, p, 0.5, 0, 0))
    for fw, fh in ((200, 72), (192, 108), (130, 100)):            # INTER_AREA asked to up-scale: OpenCV's linear area mode
        p.fit_w, p.fit_h, p.out_w, p.out_h = fw, fh, 2 * fw, fh
        got = R.finish_frame(T(L_), T(R_), T(dn), p, 0.5, 0, 0).cpu().numpy()
        assert np.array_equal(got, oracle.finish_frame(L_, R_, dn, p, 0.5, 0, 0)), (fw, fh)


@pytest.mark.parametrize("dof", [0.0, 0.7, 1.0, 2.0, 3.3, 5.0, 6.5, 7.5])
def test_dof_strengths(R, oracle, dof):
    """dof 0 (grade only), radii inside the fused fast path (<= 2.0) and beyond it: the unfused dense kernel with its tap count as a
    template parameter up to 21 taps (5.0, the GUI slider's maximum) and its run-time loop for 23 .. 31 taps (6.5: 27, 7.5: 31)."""
    _loop_eq(R, oracle, 54, 96, 2, dict(BASE, output_format="Half-SBS", output_height=54, dof_strength=dof,
                                        color_saturation=1.2, color_contrast=1.1, color_brightness=0.03))


def test_degenerate_depth_planes(R, oracle):
    sh, sw = 54, 96
    p = render_kwargs_to_params(sw, sh, **dict(BASE, output_format="Half-SBS", output_height=sh))
    f = synth.synth_frame(0, sh, sw)[0]
    planes = [np.zeros((sh, sw), np.float32), np.ones((sh, sw), np.float32), np.full((sh, sw), 0.5, np.float32),
              (np.arange(sh * sw).reshape(sh, sw) % 2).astype(np.float32), synth.synth_frame(3, sh, sw)[1]]
    R.reset_state(); R.new_clip()
    ro = oracle.RenderOracle(p); ro.new_clip()
    for i, d in enumerate(planes):
        got = R.render_frame(T(f), T(d), p).cpu().numpy()
        exp = ro.render(f, d, 0)
        assert R.last_scalars().as_dict() == ro.last.as_dict(), i
        assert np.array_equal(got, exp), i


def test_unsupported_features_fail_loudly():
    from visiondepth3d_amd.render_3d import Renderer
    r = Renderer(0)
    p = render_kwargs_to_params(96, 54, **dict(BASE, output_format="VR", output_height=54))
    p.fit_w, p.fit_h, p.out_w = 720, 800, 1440          # a VR canvas other than 1440x1600 would need format_3d_output's INTER_LINEAR resize
    with pytest.raises(Vd3dError) as e:
        r.render_frame(torch.zeros(54, 96, 3, dtype=torch.uint8).cuda(), torch.zeros(54, 96).cuda(), p)
    assert e.value.code == -4
    r.close()


def test_heal_missing_pixels_vs_oracle_and_golden(oracle):
    """a23 heal_missing_pixels (core/render_3d.py:431-459) through the C ABI: bit-exact against the reference goldens (float32
    planes), against the oracle on ragged sizes (tile edges, 1-pixel images), and on a full 1080p plane."""
    import json
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from cases import HEAL_CASES, heal_inputs
    from conftest import load_golden
    from visiondepth3d_amd.render_3d import Renderer
    r = Renderer(0)
    g = load_golden("heal.npz")
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    for i, case in enumerate(HEAL_CASES):
        warped, orig, edge, hs = heal_inputs(case)
        got = r.heal_missing_pixels(T(warped), T(orig), None if edge is None else T(edge), hs).cpu().numpy()
        assert np.array_equal(got, g[f"healed_{i}"]), i
    rng = np.random.default_rng(3)
    for (H, W) in ((1, 1), (1, 7), (5, 1), (17, 65), (16, 64), (31, 129), (1080, 1920)):
        warped = np.clip(rng.random((3, H, W)) * 0.1 + np.linspace(0, 0.9, W)[None, None, :], 0, 1).astype(np.float32)
        orig = rng.random((3, H, W)).astype(np.float32)
        edge = rng.random((1, H, W)).astype(np.float32) if (H * W) % 2 else None
        got = r.heal_missing_pixels(T(warped), T(orig), None if edge is None else T(edge), 0.5).cpu().numpy()
        exp = oracle.heal_missing_pixels(warped, orig, edge, 0.5)
        assert np.array_equal(got, exp), (H, W)
    with pytest.raises(AssertionError):
        r.heal_missing_pixels(torch.zeros(3, 4, 4), torch.zeros(3, 4, 5))
    r.close()
