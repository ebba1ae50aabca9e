"""GPU parity tests (-m gpu) for the round-1 widening rows: black-bar auto crop (core/render_3d.py:293-326,1230-1248),
fractional / mixed INTER_AREA inside pad_to_aspect_ratio (:101-131) and the VR format (:846-849).
HIP (through the C ABI) vs the CPU oracle: bit-exact; vs the reference goldens (tests/golden/widen.npz): the B2 bars."""
import os

import numpy as np
import pytest

from conftest import assert_parity, golden_json, load_golden, u8_diff_stats
from visiondepth3d_amd import synth
from visiondepth3d_amd._lib import Vd3dError
from visiondepth3d_amd.params import render_kwargs_to_params

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def R():
    from visiondepth3d_amd.render_3d import Renderer
    assert torch.cuda.is_available(), "GPU tests need a ROCm device"
    r = Renderer(0)
    yield r
    r.close()


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_detect_black_bars_vs_oracle_and_golden(R, oracle):
    g = load_golden("widen.npz")
    lf, _ = synth.letterbox_clip(4, 120, 192, 10, 14)
    got = [R.detect_black_bars(T(f)) for f in lf]
    assert got == [oracle.detect_black_bars(f) for f in lf]
    assert np.array_equal(np.array(got, np.int32), g["bars"])
    rng = np.random.default_rng(5)
    cases = {
        "no_bars": np.zeros((40, 64, 3), np.uint8) + 200,
        "all_black": np.zeros((40, 64, 3), np.uint8) + 4,
        "threshold_edge": np.zeros((9, 50, 3), np.uint8) + 10,       # gray 10 everywhere: mean == 10 is NOT > 10
        "one_bright_pixel": np.zeros((9, 50, 3), np.uint8) + 10,
        "noise_dark": rng.integers(0, 22, (37, 131, 3)).astype(np.uint8),   # row means straddle the threshold
        "single_row": rng.integers(0, 255, (1, 17, 3)).astype(np.uint8),
        "wide_4k_row": rng.integers(0, 40, (6, 3840, 3)).astype(np.uint8),
    }
    cases["one_bright_pixel"] = cases["one_bright_pixel"].copy()
    cases["one_bright_pixel"][4, 7] = 255
    for name, f in cases.items():
        assert R.detect_black_bars(T(f)) == oracle.detect_black_bars(f), name


def _loop(R, oracle, frames, depth_bgr, p):
    R.reset_state(); R.new_clip()
    ro = oracle.RenderOracle(p); ro.new_clip()
    got, exp = [], []
    for f, d in list(zip(frames, depth_bgr))[1:]:
        got.append(R.render_frame(T(f), T(d), p).cpu().numpy())
        exp.append(ro.render(f, d, 1))
        a, b = R.last_scalars().as_dict(), ro.last.as_dict()
        assert a == b, {k: (a[k], b[k]) for k in a if a[k] != b[k]}
    return np.stack(got), np.stack(exp)


def test_autocrop_loop_bit_exact(R, oracle):
    g = load_golden("widen.npz")
    sh, sw, n, kw, _ = golden_json(g, "cases_json")["autocrop_letterbox"]
    frames, depth_bgr = synth.letterbox_clip(n, sh, sw, 10, 14)
    p = render_kwargs_to_params(sw, sh, **kw)
    got, exp = _loop(R, oracle, frames, depth_bgr, p)
    assert np.array_equal(got, exp), u8_diff_stats(got, exp)
    assert_parity("autocrop_letterbox", *u8_diff_stats(got, g["autocrop_letterbox__frames"]))
    # a later clip WITHOUT auto crop on the same context reports (0, 0) again and uses the static crop
    kw2 = dict(kw); kw2["auto_crop_black_bars"] = False
    p2 = render_kwargs_to_params(sw, sh, **kw2)
    got2, exp2 = _loop(R, oracle, frames, depth_bgr, p2)
    assert np.array_equal(got2, exp2) and not np.array_equal(got2, got)
    assert R.last_scalars().crop_top == 0 and R.last_scalars().crop_bottom == 0


@pytest.mark.parametrize("fmt", ["Full-SBS", "Passive Interlaced", "Red-Cyan Anaglyph"])
def test_autocrop_other_formats(R, oracle, fmt):
    frames, depth_bgr = synth.letterbox_clip(3, 96, 160, 7, 9)
    p = render_kwargs_to_params(160, 96, output_format=fmt, output_height=90, fg_shift=8.0, mg_shift=-2.0, bg_shift=-5.0,
                                sharpness_factor=0.2, dof_strength=1.0, auto_crop_black_bars=True, preserve_original_aspect=True,
                                original_video_width=160, original_video_height=90)
    got, exp = _loop(R, oracle, frames, depth_bgr, p)
    assert np.array_equal(got, exp), u8_diff_stats(got, exp)


def test_fractional_fit_finish_vs_oracle(R, oracle):
    """finish stage with fractional / mixed INTER_AREA ratios and pad offsets (small canvases, every format that pads)."""
    rng = np.random.default_rng(11)
    H, W = 72, 128
    L = rng.integers(0, 256, (H, W, 3)).astype(np.uint8)
    Rr = rng.integers(0, 256, (H, W, 3)).astype(np.uint8)
    dn = rng.random((H, W)).astype(np.float32)
    for fmt, (fw, fh) in [("Full-SBS", (96, 60)), ("Full-SBS", (100, 48)), ("Passive Interlaced", (85, 50)),
                          ("Red-Cyan Anaglyph", (51, 40)), ("Full-SBS", (120, 70))]:
        p = render_kwargs_to_params(W, H, output_format=fmt, output_height=H, fg_shift=8.0, mg_shift=-2.0, bg_shift=-5.0,
                                    sharpness_factor=0.2, dof_strength=2.0, preserve_original_aspect=True,
                                    original_video_width=W, original_video_height=H)
        p.fit_w, p.fit_h = fw, fh
        p.out_w = 2 * fw if fmt == "Full-SBS" else fw
        p.out_h = fh
        got = R.finish_frame(T(L), T(Rr), T(dn), p, 0.4, bar_width=5, bar_side=1).cpu().numpy()
        exp = oracle.finish_frame(L, Rr, dn, p, 0.4, 5, 1)
        assert np.array_equal(got, exp), (fmt, fw, fh, u8_diff_stats(got, exp))


def test_two_by_two_fit_vector_epilogue_vs_oracle(R, oracle):
    """Round 5: the fused finishing kernel's vector epilogue also takes the 2 x 2 fit -- the GUI's own default on a 4K source (Full-SBS = 1920 x 1080
    eyes from a 3840 x 2160 warp, VisionDepth3D.py:1405-1453; OpenCV's ResizeAreaFastVec (sum + 2) >> 2).  Interior tiles take the vector path, border
    tiles the per-pixel one: both against the oracle, every format that reaches this fit, ragged sizes included."""
    rng = np.random.default_rng(31)
    for (H, W), fmt in [((156, 256), "Full-SBS"), ((208, 320), "Passive Interlaced"), ((156, 384), "Red-Cyan Anaglyph"), ((210, 300), "Full-SBS"),
                        ((2160, 3840), "Full-SBS")]:
        L = rng.integers(0, 256, (H, W, 3)).astype(np.uint8)
        Rr = rng.integers(0, 256, (H, W, 3)).astype(np.uint8)
        dn = rng.random((H // 2, W // 2)).astype(np.float32) if H > 1000 else rng.random((H, W)).astype(np.float32)
        p = render_kwargs_to_params(W, H, output_format=fmt, output_height=H, fg_shift=8.0, mg_shift=-2.0, bg_shift=-5.0,
                                    sharpness_factor=0.2, dof_strength=2.0, preserve_original_aspect=True,
                                    original_video_width=W, original_video_height=H)
        p.fit_w, p.fit_h = W // 2, H // 2
        p.out_w = W if fmt == "Full-SBS" else W // 2
        p.out_h = H // 2
        if H > 1000:
            p.eye_w, p.eye_h = W // 2, H // 2
        got = R.finish_frame(T(L), T(Rr), T(dn), p, 0.4, bar_width=7, bar_side=2).cpu().numpy()
        exp = oracle.finish_frame(L, Rr, dn, p, 0.4, 7, 2)
        assert got.shape == exp.shape and np.array_equal(got, exp), (fmt, H, W, u8_diff_stats(got, exp))


def test_fused_finish_in_front_of_any_fit_equals_the_unfused_kernels(R, oracle):
    """Round 4: a fit the fused finishing kernel does not take (fractional / up-scaling INTER_AREA, the VR canvas) no longer sends the frame to
    the unfused DOF + grade kernels: E1 runs 1:1 into a side-by-side scratch of sharpened eyes and k_sharp_mux does fit + mux only.  Both
    routes (vd3d_debug_tune(3, 0) = the unfused one) must produce the oracle's bytes: every padding format, fractional / mixed / up-scaling
    ratios, frame sizes with ragged tiles, the dense and the separable DOF order, DOF off, and a 13-tap strength that E1 refuses either way."""
    from visiondepth3d_amd import _lib
    L_ = _lib.lib()
    rng = np.random.default_rng(23)
    try:
        for (H, W), fmt, (fw, fh), dof, dense in [((72, 128), "Full-SBS", (96, 60), 2.0, True), ((72, 128), "VR", (1440, 1600), 2.0, True),
                                                   ((90, 160), "Passive Interlaced", (85, 50), 1.5, True), ((62, 100), "Red-Cyan Anaglyph", (51, 40), 2.0, False),
                                                   ((72, 128), "Full-SBS", (200, 130), 0.0, True), ((270, 480), "Full-SBS", (333, 200), 2.0, True),
                                                   ((72, 128), "Full-SBS", (96, 60), 3.0, True), ((62, 102), "Full-SBS", (60, 40), 2.0, True),
                                                   # integer fits behind a 13-tap DOF: k_sharp_fit (the fused kernel's epilogue on graded planes)
                                                   ((90, 160), "Half-SBS", (80, 90), 3.0, True), ((90, 160), "Full-SBS", (160, 90), 3.0, True),
                                                   ((90, 160), "Passive Interlaced", (160, 90), 3.0, True), ((90, 160), "Red-Cyan Anaglyph", (160, 90), 3.0, True),
                                                   ((120, 256), "Full-SBS", (64, 30), 3.0, True), ((270, 480), "Half-SBS", (240, 270), 5.0, True)]:
            Le = rng.integers(0, 256, (H, W, 3)).astype(np.uint8)
            Re = rng.integers(0, 256, (H, W, 3)).astype(np.uint8)
            dn = rng.random((H, W)).astype(np.float32)
            p = render_kwargs_to_params(W, H, output_format=fmt, output_height=H, fg_shift=8.0, mg_shift=-2.0, bg_shift=-5.0,
                                        sharpness_factor=0.2, dof_strength=dof, dof_dense_conv=dense, preserve_original_aspect=True,
                                        original_video_width=W, original_video_height=H)
            p.fit_w, p.fit_h = fw, fh
            p.out_w = 2 * fw if fmt in ("Half-SBS", "Full-SBS", "VR") else fw
            p.out_h = fh
            exp = oracle.finish_frame(Le, Re, dn, p, 0.4, 5, 2)
            outs = []
            for route in (3, 1, 0):   # bit 0: E1 in front of the fit; bit 1: k_sharp_fit (E1's epilogue) behind the unfused DOF kernels
                assert L_.vd3d_debug_tune(3, route) == 0
                outs.append(R.finish_frame(T(Le), T(Re), T(dn), p, 0.4, bar_width=5, bar_side=2).cpu().numpy())
            for name, o in zip(("all routes", "fused + fit", "unfused"), outs):
                assert np.array_equal(o, exp), (name, fmt, (H, W), (fw, fh), u8_diff_stats(o, exp))
    finally:
        L_.vd3d_debug_tune(3, 3)


def test_vr_loop_bit_exact_and_golden(R, oracle):
    g = load_golden("widen.npz")
    sh, sw, n, kw, bands = golden_json(g, "cases_json")["vr_1080"]
    frames, depths = synth.synth_clip(n, sh, sw)
    p = render_kwargs_to_params(sw, sh, **kw)
    got, exp = _loop(R, oracle, frames, [synth.depth_to_u8_bgr(d) for d in depths], p)
    assert got.shape == (2, 1600, 2880, 3)
    assert np.array_equal(got, exp), u8_diff_stats(got, exp)
    for (a, b) in bands:
        assert_parity("vr_1080", *u8_diff_stats(got[:, a:b], g[f"vr_1080__rows_{a}_{b}"]))


def test_full_sbs_720_like_upscale_loop(R, oracle):
    """Non-preserve Full-SBS from a source smaller than 1080p: fixed 1920x1080 eyes, pad_to_aspect_ratio up-scales with
    INTER_AREA (OpenCV: linear machinery with area-mode coefficients)."""
    sh, sw = 72, 128
    frames, depths = synth.synth_clip(3, sh, sw)
    p = render_kwargs_to_params(sw, sh, output_format="Full-SBS", output_height=sh, fg_shift=8.0, mg_shift=-2.0, bg_shift=-5.0,
                                sharpness_factor=0.2, dof_strength=2.0)
    assert (p.fit_w, p.fit_h, p.out_w, p.out_h) == (1920, 1080, 3840, 1080)
    got, exp = _loop(R, oracle, frames, [synth.depth_to_u8_bgr(d) for d in depths], p)
    assert np.array_equal(got, exp), u8_diff_stats(got, exp)
    assert got[:, :, 1900:1920].any() and got[:, 0].any() and got[:, 1079].any()     # the picture fills the 1920x1080 eye canvas


def test_pinned_ring_clip_equals_sequential(R, oracle):
    """frame_io.render_clip_pipelined (pinned staging, copy streams) returns the same bytes as frame-by-frame rendering."""
    from visiondepth3d_amd.frame_io import render_clip_pipelined
    sh, sw, n = 108, 192, 9
    frames, depths = synth.synth_clip(n, sh, sw)
    dep = [synth.depth_to_u8_bgr(d) for d in depths]
    p = render_kwargs_to_params(sw, sh, output_format="Half-SBS", output_height=sh, fg_shift=8.0, mg_shift=-2.0, bg_shift=-5.0,
                                sharpness_factor=0.2, dof_strength=2.0, feather_strength=10.0, blur_ksize=9,
                                use_subject_tracking=True, use_floating_window=True)
    R.reset_state(); R.new_clip()
    seq = [R.render_frame(T(f), T(d), p).cpu().numpy() for f, d in list(zip(frames, dep))[1:]]
    R.reset_state()
    got = list(render_clip_pipelined(R, frames, dep, p, depth=3))
    assert len(got) == len(seq) == n - 1
    for a, b in zip(got, seq):
        assert np.array_equal(a, b)


def test_autocrop_in_sharded_steps_equals_sequential(R):
    """auto_crop_black_bars inside chunk-sharded steps: every owner detects the bars of its own frames (P0) and ingests them with
    that rectangle -- no exchange is needed, only owners ingest.  Two emulated ranks (one context each) vs the sequential render."""
    from shard_emul import Emu
    from visiondepth3d_amd.render_3d import Renderer
    from visiondepth3d_amd.sharded import ChunkSharder, HipChunkBackend
    sh, sw, G, B, steps = 120, 192, 2, 2, 2
    n = steps * G * B
    frames, depth_bgr = synth.letterbox_clip(n, sh, sw, 10, 14)
    gray = [d[..., 0].copy() for d in depth_bgr]
    p = render_kwargs_to_params(sw, sh, output_format="Half-SBS", output_height=108, fg_shift=10.0, mg_shift=-2.5, bg_shift=-5.0,
                                sharpness_factor=0.15, dof_strength=2.0, feather_strength=10.0, blur_ksize=9,
                                use_subject_tracking=True, use_floating_window=True, auto_crop_black_bars=True)
    ft, dt = [T(f) for f in frames], [T(d) for d in gray]
    R.reset_state(); R.new_clip()
    seq = [R.render_frame(f, d, p).cpu().numpy() for f, d in zip(ft, dt)]
    st_seq = R.export_state().as_dict()
    ranks = [Renderer(0) for _ in range(G)]
    shd = []
    for g_, rr in enumerate(ranks):
        rr.reset_state(); rr.new_clip()
        shd.append(ChunkSharder(HipChunkBackend(rr, p), g_, G, B))
    assert all(s_.auto_crop for s_ in shd)
    got = Emu(shd).run_clip(ft, dt, B)
    for t in range(n):
        assert np.array_equal(got[t].cpu().numpy(), seq[t]), t
    assert all(rr.export_state().as_dict() == st_seq for rr in ranks)
    crops = torch.cat([s_.crops[g_ * B:(g_ + 1) * B] for g_, s_ in enumerate(shd)]).cpu().numpy()   # the last step's rectangles
    assert (crops[:, 1] >= 10).all() and (crops[:, 3] <= 96).all()      # the letterbox rows are gone from every frame's rectangle
    for rr in ranks:
        rr.close()


def test_preview_visualisers_vs_oracle_and_golden(R, oracle):
    from visiondepth3d_amd.preview_utils import PREVIEW_TYPES, generate_preview_image, preview_image
    g = load_golden("previews.npz")
    for tag, (h, w) in {"even": (54, 96), "odd": (37, 75), "hd": (1080, 1920)}.items():
        left, right = synth.synth_frame(1, h, w)[0], synth.synth_frame(2, h, w)[0]
        for pt in PREVIEW_TYPES:
            got = preview_image(R, pt, T(left), T(right)).cpu().numpy()
            assert np.array_equal(got, oracle.preview_image(pt, left, right)), (tag, pt)
            if tag != "hd":
                assert np.array_equal(got, g[f"{tag}__{pt}"]), (tag, pt)
    assert generate_preview_image("no such preview", left, right, None, w, h) is None      # the reference returns None as well


def test_preview_arrows_vs_oracle(R, oracle):
    """"Overlay Arrows" (core/preview_utils.py:74-82): the closed-form HIP kernel against the oracle's LineIterator-style restatement, on
    random shift maps (every arrow direction and length, arrows crossing each other and the image border)."""
    from visiondepth3d_amd.preview_utils import generate_preview_image, preview_arrows
    for seed, (h, w, amp) in enumerate([(54, 96, 1.5), (37, 75, 6.0), (1080, 1920, 3.0), (200, 300, 40.0)]):
        rng = np.random.default_rng(seed)
        left = synth.synth_frame(seed, h, w)[0]
        shift = (rng.standard_normal((1, h, w)) * amp).astype(np.float32)
        shift[0, 0, 0] = np.nan
        got = preview_arrows(R, T(left), T(shift)).cpu().numpy()
        assert np.array_equal(got, oracle.preview_arrows(left, shift)), (h, w)
    out = generate_preview_image("Overlay Arrows", left, left, torch.from_numpy(shift), w, h)
    assert np.array_equal(out, got)


def test_preview_heatmaps_vs_oracle_and_golden(R, oracle):
    """The colour-mapped previews (core/preview_utils.py:42-66): index plane on device, table from the caller.  All four types against
    the oracle; the two whose arithmetic is plain numpy also against the reference fixture (tests/golden/previews_heat.npz)."""
    from visiondepth3d_amd import preview_utils as PU
    g = load_golden("previews_heat.npz")
    luts = {"JET": g["lut_JET"], "BONE": g["lut_BONE"]}
    for tag in ("even", "odd"):
        shift = g[f"{tag}__shift"]
        for pt, (kind, cmap) in PU.HEATMAP_TYPES.items():
            got = PU.preview_heatmap(R, pt, T(shift), lut=luts[cmap]).cpu().numpy()
            assert np.array_equal(got, oracle.preview_heatmap(kind, shift, luts[cmap])), (tag, pt)
            if f"{tag}__{pt}" in g.files:
                assert np.array_equal(got, g[f"{tag}__{pt}"]), (tag, pt)
    # frame-size plane, constant plane (cv2.normalize: scale 0), the reference-signature entry with a registered table
    rng = np.random.default_rng(3)
    big = (rng.standard_normal((1, 1080, 1920)) * 4).astype(np.float32)
    for pt, (kind, cmap) in PU.HEATMAP_TYPES.items():
        assert np.array_equal(PU.preview_heatmap(R, pt, T(big), lut=luts[cmap]).cpu().numpy(), oracle.preview_heatmap(kind, big, luts[cmap])), pt
    flat = np.full((1, 20, 30), 1.25, np.float32)
    assert np.array_equal(PU.preview_heatmap(R, "Shift Heatmap", T(flat), lut=luts["JET"]).cpu().numpy(),
                          oracle.preview_heatmap(0, flat, luts["JET"]))
    saved = dict(PU._COLORMAPS)
    try:
        PU._COLORMAPS.clear()
        try:
            import cv2  # noqa: F401
        except ImportError:
            with pytest.raises(NotImplementedError):
                PU.generate_preview_image("Feather Mask", None, None, torch.from_numpy(flat), 30, 20)
        PU.register_colormap("bone", luts["BONE"])
        out = PU.generate_preview_image("Feather Mask", None, None, torch.from_numpy(flat), 30, 20)
        assert np.array_equal(out, oracle.preview_heatmap(3, flat, luts["BONE"]))
    finally:
        PU._COLORMAPS.clear()
        PU._COLORMAPS.update(saved)


# ---- skip_blank_frames (core/render_3d.py:1046-1060,1278-1281; tests/golden/blank.npz) ---------------------------------
def test_blank_frame_loops(R, oracle):
    """vd3d_render_frame_blank inside real loops: HIP vs oracle bit-exact (pixels and every reported scalar, blank or not),
    blank frames bit-exact vs the reference golden, the others within the B2 bar."""
    g = load_golden("blank.npz")
    for name, (sh, sw, n, blank, kw) in golden_json(g, "cases_json").items():
        frames, depths = synth.synth_clip(n, sh, sw)
        p = render_kwargs_to_params(sw, sh, **kw)
        R.reset_state(); R.new_clip()
        ro = oracle.RenderOracle(p); ro.new_clip()
        for idx, (f, d) in enumerate(list(zip(frames, depths))[1:]):
            d8 = synth.depth_to_u8_bgr(d)
            b = idx in blank
            got = R.render_frame(T(f), T(d8), p, blank=b).cpu().numpy()
            exp = ro.render(f, d8, 1, blank=b)
            assert np.array_equal(got, exp), (name, idx, b, u8_diff_stats(got, exp))
            sa, sb = R.last_scalars().as_dict(), ro.last.as_dict()
            assert sa == sb, (name, idx, {k: (sa[k], sb[k]) for k in sa if sa[k] != sb[k]})
            ref = g[f"{name}__frames"][idx]
            if b:
                assert np.array_equal(got, ref), (name, idx)
            else:
                assert_parity(name, *u8_diff_stats(got, ref))
        assert R.export_state().as_dict() == ro.state.as_dict(), name


def test_render_clip_batched_steps_equal_frame_by_frame(R):
    """render_pairs in steps of `batch` frames (the default of the file-to-file loop: sharded.ChunkSharder at world 1, two slot sets, two pixel
    streams) == one vd3d_render_frame per pair (batch=1): 11 frames in steps of 3 (a partial last step), a blank frame, float32 depth and the
    uint8 depth-video format; the renderer is back in sequential mode afterwards."""
    from visiondepth3d_amd.render_3d import render_clip
    sh, sw, n = 216, 384, 12
    kw = dict(output_format="Half-SBS", output_height=sh, fg_shift=10.0, mg_shift=-2.5, bg_shift=-5.0, sharpness_factor=0.15, dof_strength=2.0,
              feather_strength=10.0, blur_ksize=9, use_subject_tracking=True, use_floating_window=True, skip_blank_frames=True)
    frames, depths = synth.synth_clip(n, sh, sw)
    for dl in (depths, [synth.depth_to_u8_bgr(d) for d in depths]):
        R.reset_state()
        seq = list(render_clip(frames, dl, renderer=R, blank_frames=[4], batch=1, **kw))
        st_seq = R.export_state().as_dict()
        for b in (3, 8):
            R.reset_state()
            got = list(render_clip(frames, dl, renderer=R, blank_frames=[4], batch=b, **kw))
            assert len(got) == len(seq) == n - 1
            assert all(np.array_equal(a, c) for a, c in zip(got, seq)), b
            assert R.export_state().as_dict() == st_seq
    f, d = T(frames[0]), T(depths[0])
    p = render_kwargs_to_params(sw, sh, **{k: v for k, v in kw.items()})
    R.render_frame(f, d, p)   # sequential entry point still works after the batched loops


def test_render_clip_blank_list(R, oracle):
    """render_clip(skip_blank_frames=True, blank_frames=[...]) == the per-frame calls; without the flag the list is ignored."""
    from visiondepth3d_amd.render_3d import render_clip
    g = load_golden("blank.npz")
    sh, sw, n, blank, kw = golden_json(g, "cases_json")["blank_half_sbs"]
    frames, depths = synth.synth_clip(n, sh, sw)
    d8 = [synth.depth_to_u8_bgr(d) for d in depths]
    R.reset_state()
    outs = np.stack(list(render_clip(frames, d8, renderer=R, blank_frames=blank, **kw)))
    mxs = [u8_diff_stats(outs[i], g["blank_half_sbs__frames"][i])[0] for i in range(len(outs))]
    assert all(m == 0 for i, m in enumerate(mxs) if i in blank) and max(mxs) <= 8
    # start_frame_idx shifts which loop iterations hit the list (:1063,1278)
    R.reset_state()
    outs2 = np.stack(list(render_clip(frames, d8, renderer=R, blank_frames=[b + 100 for b in blank], start_frame_idx=100, **kw)))
    assert np.array_equal(outs, outs2)
    kw2 = dict(kw); kw2["skip_blank_frames"] = False
    R.reset_state()
    outs3 = np.stack(list(render_clip(frames, d8, renderer=R, blank_frames=blank, **kw2)))
    assert not np.array_equal(outs3[blank[0]], outs[blank[0]])


def test_format_3d_output_and_linear_resize(R, oracle):
    """format_3d_output (core/render_3d.py:837-860) as an entry point of its own (vd3d_format_3d_output) for every format, incl. VR eyes that are not
    1440x1600 (cv2.resize INTER_LINEAR first, :846-849; unpinned OpenCV arithmetic: HIP == oracle bit for bit), and the linear resize alone for
    up- and down-scaling shapes; the module-level format_3d_output / generate_anaglyph_3d take and return NumPy arrays like the reference's."""
    from visiondepth3d_amd import render_3d as r3
    rng = np.random.default_rng(5)
    L = rng.integers(0, 256, (90, 160, 3), dtype=np.uint8)
    Rr = rng.integers(0, 256, (90, 160, 3), dtype=np.uint8)
    for name, code in (("Half-SBS", 0), ("Full-SBS", 1), ("VR", 2), ("Red-Cyan Anaglyph", 3), ("Passive Interlaced", 4)):
        got = R.format_3d_output(T(L), T(Rr), name).cpu().numpy()
        exp = oracle.format_output(L, Rr, code)
        assert got.shape == exp.shape and np.array_equal(got, exp), name
    assert np.array_equal(R.format_3d_output(T(L), T(Rr), "no such format").cpu().numpy(), np.hstack((L, Rr)))   # :860 fallback
    # eye sizes at which the render loop's pad_to_aspect_ratio would truncate int(w / h * h) to w - 1 (about one size in twenty): format_3d_output has
    # no such step -- np.hstack / row interleave / the anaglyph mix of the eyes as they are (the last column came out black)
    for (eh, ew) in ((7, 61), (7, 115), (804, 1920), (800, 854)):
        assert int((ew / eh) * eh) == ew - 1, (eh, ew)
        Lt = rng.integers(1, 256, (eh, ew, 3), dtype=np.uint8)
        Rt = rng.integers(1, 256, (eh, ew, 3), dtype=np.uint8)
        for name, code in (("Half-SBS", 0), ("Full-SBS", 1), ("Red-Cyan Anaglyph", 3), ("Passive Interlaced", 4)):
            got = R.format_3d_output(T(Lt), T(Rt), name).cpu().numpy()
            exp = oracle.format_output(Lt, Rt, code)
            assert got.shape == exp.shape and np.array_equal(got, exp), (name, eh, ew)
        assert np.array_equal(R.format_3d_output(T(Lt), T(Rt), "Full-SBS").cpu().numpy(), np.hstack((Lt, Rt)))
    Lv = rng.integers(0, 256, (1600, 1440, 3), dtype=np.uint8)
    assert np.array_equal(R.format_3d_output(T(Lv), T(Lv[::-1].copy()), "VR").cpu().numpy(), np.hstack((Lv, Lv[::-1])))   # identity resize
    for (sh, sw), (dh, dw) in (((90, 160), (1600, 1440)), ((270, 480), (135, 240)), ((37, 53), (80, 31)), ((64, 64), (64, 64)), ((5, 7), (11, 3))):
        src = rng.integers(0, 256, (sh, sw, 3), dtype=np.uint8)
        assert np.array_equal(R.resize_linear_u8(T(src), dh, dw).cpu().numpy(), oracle.resize_linear_u8(src, dh, dw)), (sh, sw, dh, dw)
    flat = np.full((20, 30, 3), 77, np.uint8)
    assert np.all(oracle.resize_linear_u8(flat, 50, 41) == 77)   # a constant image stays constant (weights sum to 2048)
    prev = r3._default
    r3._default = R
    try:
        assert np.array_equal(r3.format_3d_output(L, Rr, "Passive Interlaced"), oracle.format_output(L, Rr, 4))
        assert np.array_equal(r3.generate_anaglyph_3d(L, Rr), oracle.format_output(L, Rr, 3))
    finally:
        r3._default = prev


@pytest.mark.reference_threads(8)   # the fixtures' reference process ran torch with 8 threads: NO extension keyword, no override -- the shim's own default
def test_render_sbs_3d_end_to_end_equals_the_reference_loop(R, monkeypatch):
    """B2's Python face on the GPU: visiondepth3d_amd.video_io.render_sbs_3d with the in-memory video backend (tests/golden/ref_stubs.py -- the
    same fake cv2 the reference's loop ran on when the fixtures were generated) and the real renderer, i.e. the batched step path of render_pairs
    (RENDER_BATCH frames per step, two pixel streams): the frames it writes must equal the frames the REFERENCE's render_sbs_3d wrote for the
    same clips (tests/golden/render_loop.npz, dof_levels.npz), for steps of 8, 2 and 1 frames."""
    import sys, threading
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import ref_stubs
    from visiondepth3d_amd import video_io
    monkeypatch.setattr(video_io, "video_backend", ref_stubs)
    for gname, names in (("render_loop.npz", ("half_sbs_cli", "full_sbs_preserve", "interlaced", "anaglyph_43crop")), ("dof_levels.npz", ("half_sbs_dof3p0",))):
        g = load_golden(gname)
        cases = golden_json(g, "cases_json")
        for name in names:
            sh, sw, n, kw = cases[name]
            frames, depths = synth.synth_clip(n, sh, sw)
            for batch in (8, 2, 1):
                monkeypatch.setattr(video_io, "RENDER_BATCH", batch)
                ref_stubs._Clip.clips["in.mp4"] = frames
                ref_stubs._Clip.clips["depth.mp4"] = [synth.depth_to_u8_bgr(d) for d in depths]
                ref_stubs._Clip.written.pop("out.avi", None)
                R.reset_state()
                args = dict(input_path="in.mp4", depth_path="depth.mp4", output_path="out.avi", selected_codec="XVID", fps=24.0, output_width=sw,
                            selected_aspect_ratio="Default (16:9)", aspect_ratios={"Default (16:9)": 16 / 9},
                            suspend_flag=threading.Event(), cancel_flag=threading.Event())
                args.update(kw)
                video_io.render_sbs_3d(**args, renderer=R)
                got = np.stack(ref_stubs._Clip.written["out.avi"])
                exp = g[f"{name}__frames"]
                assert got.shape == exp.shape and np.array_equal(got, exp), (name, batch, u8_diff_stats(got, exp) if got.shape == exp.shape else got.shape)


def _run_shell(video_io, ref_stubs, R, sh, sw, n, kw):
    import threading
    frames, depths = synth.synth_clip(n, sh, sw)
    ref_stubs._Clip.clips["in.mp4"] = frames
    ref_stubs._Clip.clips["depth.mp4"] = [synth.depth_to_u8_bgr(d) for d in depths]
    ref_stubs._Clip.written.pop("out.avi", None)
    R.reset_state()
    args = dict(input_path="in.mp4", depth_path="depth.mp4", output_path="out.avi", selected_codec="XVID", fps=24.0, output_width=sw,
                selected_aspect_ratio="Default (16:9)", aspect_ratios={"Default (16:9)": 16 / 9}, suspend_flag=threading.Event(), cancel_flag=threading.Event())
    args.update(kw)
    video_io.render_sbs_3d(**args, renderer=R)
    return ref_stubs._Clip.written["out.avi"]


def test_drop_in_default_is_the_reference_processes_thread_count(R, monkeypatch):
    """Round 6.  The 51-parameter entry with NO extension keyword and no environment override reproduces the live reference's frames at sizes
    where its arithmetic depends on torch.get_num_threads() (tests/golden/aten_any_size.npz: odd sizes, thumbnails, cropped sources, each rendered by the reference
    with a fixed thread count) whenever the calling process runs torch with that many threads -- the shim IS running in the reference's process.  Then the same
    through VD3D_ATEN_THREADS from a process with another thread count, and the proof that the default matters: with the thread-independent arithmetic
    (VD3D_ATEN_THREADS=0) at least three of those frames differ."""
    import hashlib, json, sys
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import ref_stubs
    from visiondepth3d_amd import video_io
    monkeypatch.setattr(video_io, "video_backend", ref_stubs)
    g = load_golden("aten_any_size.npz")
    cases = json.loads(bytes(g["cases_json"]).decode())
    prev = torch.get_num_threads()

    def frames_match(name, outs):
        ok = len(outs) == 3
        for i, out in enumerate(outs):
            ok = ok and tuple(out.shape) == tuple(int(v) for v in g[f"{name}__shape"])
            ok = ok and hashlib.sha256(np.ascontiguousarray(out).tobytes()).hexdigest()[:16] == bytes(g[f"{name}__sha_{i}"]).decode()
        return ok
    try:
        monkeypatch.delenv("VD3D_ATEN_THREADS", raising=False)
        for name, (sh, sw, kw, threads) in cases.items():
            torch.set_num_threads(threads)
            for batch in (8, 1):
                monkeypatch.setattr(video_io, "RENDER_BATCH", batch)
                assert frames_match(name, _run_shell(video_io, ref_stubs, R, sh, sw, 4, kw)), (name, batch, "default = torch.get_num_threads()")
        monkeypatch.setattr(video_io, "RENDER_BATCH", 8)
        torch.set_num_threads(5)   # a host with another core count renders a clip the way a 1- / 4-thread reference did
        for name in ("any1_t4", "any2_t1", "any15_t1"):
            sh, sw, kw, threads = cases[name]
            monkeypatch.setenv("VD3D_ATEN_THREADS", str(threads))
            assert frames_match(name, _run_shell(video_io, ref_stubs, R, sh, sw, 4, kw)), (name, "VD3D_ATEN_THREADS")
        monkeypatch.setenv("VD3D_ATEN_THREADS", "0")
        differing = sum(not frames_match(name, _run_shell(video_io, ref_stubs, R, *cases[name][:2], 4, cases[name][2])) for name in ("any1_t4", "any2_t1", "any15_t1"))
        assert differing >= 1
    finally:
        torch.set_num_threads(prev)


def test_pixel_shift_cuda_shim_default_is_the_reference_processes_thread_count(monkeypatch):
    """The B1 shim with the reference's own keywords only (core/render_3d.py:561-585): small odd planes rendered by the live reference with 1 .. 8 torch threads
    (tests/golden/pixel_shift_small_planes.npz) -- float32 shift map, both eyes and the module-level tracker equal the reference's when this process runs torch with
    the fixture's thread count."""
    import torch
    from visiondepth3d_amd import render_3d as r3
    g = load_golden("pixel_shift_small_planes.npz")
    meta = golden_json(g, "meta_json")
    prev = torch.get_num_threads()
    monkeypatch.delenv("VD3D_ATEN_THREADS", raising=False)
    try:
        for key, m in meta.items():
            kw = dict(m["kw"])
            threads = kw.pop("aten_threads")
            torch.set_num_threads(threads)
            bgr, d = synth.synth_frame(m["frame_idx"], m["ih"], m["iw"])
            r3.default_renderer().reset_state()
            L, Rr, S = r3.pixel_shift_cuda(r3.frame_to_tensor(bgr), T(d[None]), m["W"], m["H"], m["fg"], m["mg"], m["bg"], **kw)
            assert r3.default_renderer().export_state().fw_prev_offset == float(g[key + "__prev_offset"]), key
            assert np.array_equal(S.numpy().view(np.uint32), g[key + "__S"].view(np.uint32)), key
            assert np.array_equal(L, g[key + "__L"]) and np.array_equal(Rr, g[key + "__R"]), (key, u8_diff_stats(L, g[key + "__L"]))
    finally:
        torch.set_num_threads(prev)
