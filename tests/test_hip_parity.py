"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against the CPU oracle on the same
seeded inputs, against the committed golden fixtures, and -- at full 1080p/4K size -- through
size-independent properties and an independent exact check.

Bars: uint8 / index / order-statistic results are BIT-EXACT against the oracle (both sides follow one
arithmetic contract, DESIGN.md "Numerics"); against the reference goldens the bars are those of
tests/test_oracle_vs_golden.py (<= 1 LSB per channel at every stage boundary)."""
import numpy as np
import pytest

from conftest import assert_parity, golden_json, load_golden, u8_diff_stats
from visiondepth3d_amd import synth
from visiondepth3d_amd._abi import ShiftParams, State
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


# ------------------------------------------------------------------------------------------ reductions
def test_quantiles_exact_vs_oracle(R, oracle):
    rng = np.random.default_rng(7)
    for n in (1000, 36864, 518400, 2073600):
        v = rng.random(n).astype(np.float32)
        qs = [0.02, 0.98, 0.05, 0.95, 0.5]
        got = R.quantiles(T(v), qs)
        exp = [oracle.quantile(v, q) for q in qs]
        assert got == exp, (n, got, exp)


def test_quantiles_adversarial(R, oracle):
    n = 96 * 54
    cases = {
        "constant": np.full(n, 0.4, np.float32),
        "zeros_and_ones": (np.arange(n) % 3 == 0).astype(np.float32),
        "u8_levels": (np.random.default_rng(1).integers(0, 256, n) / 255).astype(np.float32),
        "two_values_close": np.where(np.arange(n) % 2 == 0, np.float32(0.3), np.nextafter(np.float32(0.3), np.float32(1))).astype(np.float32),
        "tiny": (np.random.default_rng(2).random(n) * 1e-30).astype(np.float32),
    }
    for name, v in cases.items():
        got = R.quantiles(T(v), [0.02, 0.98])
        exp = [oracle.quantile(v, 0.02), oracle.quantile(v, 0.98)]
        assert got == exp, name


def test_quantile_4k_vs_torch_sort(R):
    """Full-size independent check: torch.quantile on the GPU is sort-based and exact."""
    g = torch.Generator(device="cuda").manual_seed(3)
    v = torch.rand(2160 * 3840, device="cuda", generator=g)
    qs = [0.05, 0.95]
    got = R.quantiles(v, qs)
    exp = [float(torch.quantile(v, q)) for q in qs]
    assert got == exp


def test_subject_depth_edge_cases(R, oracle):
    g = load_golden("helpers.npz")
    for k in ("ramp", "few_valid", "ones_edge", "two_peaks", "const_mid"):
        p = g[f"subj_in__{k}"]
        assert np.float32(R.subject_depth(T(p))) == g[f"subj_out__{k}"], k
    p = synth.synth_frame(1, 1080, 1920)[1]
    assert R.subject_depth(T(p)) == oracle.subject_depth(p)


def test_depth_handoff_bit_exact_vs_oracle_and_close_to_torch(R, oracle):
    """a24: HIP == oracle exactly; both within 1 LSB (rare) of torch's bicubic + numpy min-max/truncate."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(5)
    pred = (torch.randn(3, 74, 132, generator=g) * 3 + 5).float()
    pred[2] = 1.25  # flat prediction -> the reference writes zeros
    H, W = 216, 384
    got = R.depth_handoff(pred.cuda(), H, W).cpu().numpy()
    inv = R.depth_handoff(pred.cuda(), H, W, invert=True).cpu().numpy()
    for b in range(3):
        exp = oracle.depth_handoff(pred[b].numpy(), H, W)
        assert np.array_equal(got[b], exp), b
        assert np.array_equal(inv[b], 255 - exp), b
    ref = F.interpolate(pred[:2, None], size=(H, W), mode="bicubic", align_corners=False)[:, 0].numpy()
    for b in range(2):
        mn, mx = ref[b].min(), ref[b].max()
        u8 = (((ref[b] - mn) / (mx - mn + np.float32(1e-6))) * 255).astype(np.uint8)
        mx_d, frac, _ = u8_diff_stats(got[b], u8)
        assert mx_d <= 1 and frac < 1e-3
    same = R.depth_handoff(pred.cuda(), 74, 132).cpu().numpy()  # identity size: no resampling
    assert np.array_equal(same[0], oracle.depth_handoff(pred[0].numpy(), 74, 132))
    assert not same[2].any()  # exactly flat prediction (max - min < 1e-6) -> the reference writes zeros (:603-606)


# ------------------------------------------------------------------------------------------ B1
def _shift_cases():
    g = load_golden("pixel_shift_cases.npz")
    return g, golden_json(g, "meta_json")


def test_pixel_shift_bit_exact_vs_oracle_and_golden(R, oracle):
    g, meta = _shift_cases()
    for key, m in meta.items():
        bgr, d = synth.synth_frame(m["frame_idx"], m["ih"], m["iw"])
        ft = oracle.frame_to_tensor(bgr)
        p = ShiftParams.defaults(m["fg"], m["mg"], m["bg"], **m["kw"])
        st = State()
        o = oracle.pixel_shift(ft, d[None], m["W"], m["H"], p, st, want_shift=True, want_dshaped=True)
        R.reset_state()
        L, Rr, S = R.pixel_shift(T(ft), T(d[None]), m["W"], m["H"], p, want_shift=True)
        planes = R.debug_planes(m["H"], m["W"])
        sc = R.last_scalars()
        assert (sc.s0, sc.q05, sc.q95, sc.s1) == tuple(np.float32(o["dbg"][k]) for k in ("s0", "q05", "q95", "s1")), key
        assert np.array_equal(planes["D"].cpu().numpy(), o["dshaped"][0]), key
        assert np.array_equal(S.cpu().numpy(), o["shift"]), key
        assert np.array_equal(L.cpu().numpy(), o["left"]), key
        assert np.array_equal(Rr.cpu().numpy(), o["right"]), key
        assert R.export_state().fw_prev_offset == st.fw_prev_offset == float(g[key + "__prev_offset"]), key
        assert np.array_equal(S.cpu().numpy(), g[key + "__S"]), key       # == the REFERENCE's shift map and eyes, bit for bit
        for eye, arr in (("L", L), ("R", Rr)):
            assert np.array_equal(arr.cpu().numpy(), g[f"{key}__{eye}"]), (key, eye, u8_diff_stats(arr.cpu().numpy(), g[f"{key}__{eye}"]))


def test_pixel_shift_small_planes_hip_vs_reference_fixture(R, oracle):
    """Round 5: eight small, odd planes rendered by the live reference with 1 .. 8 torch threads (tests/golden/pixel_shift_small_planes.npz) through the C ABI in the N-thread
    ATen mode: float32 shift map, both eyes and the tracker equal the reference's -- scalar tails (k_chain_shape<true> / k_shift<true>), the premultiplied-weight bilinear
    kernel ahead of W1, torch.quantile's fused lerp in the select chain's scalar stage."""
    from test_oracle_vs_golden import small_planes_check

    def run(bgr, d, W, H, p):
        R.reset_state()
        L, Rr, S = R.pixel_shift(T(oracle.frame_to_tensor(bgr)), T(d[None]), W, H, p, want_shift=True)
        return S.cpu().numpy(), L.cpu().numpy(), Rr.cpu().numpy(), R.export_state().fw_prev_offset
    assert small_planes_check(run) == 8


def test_pixel_shift_cuda_signature_and_singleton(oracle):
    """Reference-shaped call: host arrays out, module-level tracker persists across calls."""
    from visiondepth3d_amd import render_3d as r3
    g = load_golden("pixel_shift_cases.npz")
    r3.default_renderer().reset_state()
    for idx in range(3):
        bgr, d = synth.synth_frame(idx, 96, 160)
        out = r3.pixel_shift_cuda(r3.frame_to_tensor(bgr), T(d[None]), 160, 96, 10.0, -2.5, -5.0, return_shift_map=False)
        assert len(out) == 2 and out[0].dtype == np.uint8 and out[0].shape == (96, 160, 3)
        assert r3.default_renderer().export_state().fw_prev_offset == float(g["seq_prev_offsets"][idx])
        for arr, k in zip(out, ("L", "R")):
            assert np.array_equal(arr, g[f"seq{idx}__{k}"]), (idx, k)
    L, Rr, S = r3.pixel_shift_cuda(r3.frame_to_tensor(bgr), T(d[None]), 160, 96, 10.0, -2.5, -5.0)
    assert S.shape == (1, 96, 160) and S.device.type == "cpu"
    with pytest.raises(AssertionError):
        r3.pixel_shift_cuda(r3.frame_to_tensor(bgr), T(d[None, :50]), 160, 96, 10.0, -2.5, -5.0)
    with pytest.raises(TypeError):
        r3.pixel_shift_cuda(r3.frame_to_tensor(bgr), T(d[None]), 160, 96, 10.0, -2.5, -5.0, no_such_kw=1)


def test_kat_appendix_a(R, oracle):
    g = load_golden("kat_appendix_a.npz")
    y, x = np.mgrid[0:144, 0:256]
    bgr = np.stack([((3 * x + 5 * y + k) % 256) for k in (0, 7, 14)], axis=2).astype(np.uint8)
    d = (((4 * x + 3 * y) % 256) / 255).astype(np.float32)[None]
    ft = oracle.frame_to_tensor(bgr)
    R.reset_state()
    L, Rr, S = R.pixel_shift(T(ft), T(d), 256, 144, ShiftParams.defaults(10, -2.5, -5), want_shift=True)
    sc = R.last_scalars()
    assert np.float32(sc.s0) == g["s0"] and np.float32(sc.q05) == g["q05"] and np.float32(sc.q95) == g["q95"]
    assert np.float32(sc.s1) == g["s1"]
    assert R.export_state().fw_prev_offset == float(g["trk_prev_offset"])
    assert np.array_equal(S.cpu().numpy(), g["trk_S"])
    for arr, k in ((L, "trk_L"), (Rr, "trk_R")):
        assert np.array_equal(arr.cpu().numpy(), g[k]), (k, u8_diff_stats(arr.cpu().numpy(), g[k]))


# ------------------------------------------------------------------------------------------ B2
def _run_loop_hip(R, sh, sw, n, kw, depth_as="bgr_u8"):
    frames, depths = synth.synth_clip(n, sh, sw)
    p = render_kwargs_to_params(sw, sh, **kw)
    R.new_clip()
    outs, scal = [], []
    for f, d in list(zip(frames, depths))[1:]:
        dd = synth.depth_to_u8_bgr(d) if depth_as == "bgr_u8" else d
        outs.append(R.render_frame(T(f), T(dd), p).cpu().numpy())
        scal.append(R.last_scalars().as_dict())
    return np.stack(outs), scal, p


def _run_loop_oracle(oracle, sh, sw, n, kw, state=None, depth_as="bgr_u8"):
    frames, depths = synth.synth_clip(n, sh, sw)
    p = render_kwargs_to_params(sw, sh, **kw)
    ro = oracle.RenderOracle(p, state)
    ro.new_clip()
    outs, scal = [], []
    for f, d in list(zip(frames, depths))[1:]:
        if depth_as == "bgr_u8":
            outs.append(ro.render(f, synth.depth_to_u8_bgr(d), 1))
        else:
            outs.append(ro.render(f, d, 0))
        scal.append(ro.last.as_dict())
    return np.stack(outs), scal, ro


def test_render_loop_bit_exact_vs_oracle_and_golden(R, oracle):
    g = load_golden("render_loop.npz")
    cases = golden_json(g, "cases_json")
    for name, (sh, sw, n, kw) in cases.items():
        R.reset_state()
        got, sg, _ = _run_loop_hip(R, sh, sw, n, kw)
        exp, se, _ = _run_loop_oracle(oracle, sh, sw, n, kw)
        for a, b in zip(sg, se):
            assert a == b, (name, {k: (a[k], b[k]) for k in a if a[k] != b[k]})
        assert np.array_equal(got, exp), (name, u8_diff_stats(got, exp))
        assert_parity(name, *u8_diff_stats(got, g[f"{name}__frames"]))


def test_dof_strength_fixtures_bit_exact(R, oracle):
    """DOF strengths 0.7 / 2.1 / 2.6 / 3.0 / 4.2 / 5.0 (tests/golden/dof_levels.npz, generated by the live reference): levels whose vsExp value is
    not the rounded exponential, Gaussians of up to 21 taps (beyond the fused finishing kernel's 9: the unfused DOF kernels), anaglyph."""
    g = load_golden("dof_levels.npz")
    for name, (sh, sw, n, kw) in golden_json(g, "cases_json").items():
        R.reset_state()
        got, _, _ = _run_loop_hip(R, sh, sw, n, kw)
        exp, _, _ = _run_loop_oracle(oracle, sh, sw, n, kw)
        assert np.array_equal(got, exp), (name, u8_diff_stats(got, exp))
        assert np.array_equal(got, g[f"{name}__frames"]), (name, u8_diff_stats(got, g[f"{name}__frames"]))


def test_large_shifts_reach_every_hh_chunk(R, oracle):
    """W1 builds only the 64-column chunks of its pre-interpolated rows that the tile's largest |shift| can reach (round 4): a wide frame with layer
    shifts that run into a +-8 % clamp (77 px at 960: four chunks per tile) next to flat regions (one or two chunks) must still equal the oracle."""
    sh, sw = 540, 960
    kw = dict(output_format="Half-SBS", output_height=sh, fg_shift=90.0, mg_shift=-20.0, bg_shift=-80.0, sharpness_factor=0.15, dof_strength=2.0,
              feather_strength=10.0, blur_ksize=9, use_subject_tracking=True, use_floating_window=True, max_pixel_shift_percent=0.08)
    p = render_kwargs_to_params(sw, sh, **kw)
    assert (p.warp_w, p.warp_h) == (sw, sh)
    frames, depths = synth.synth_clip(2, sh, sw)
    R.reset_state(); R.new_clip()
    ro = oracle.RenderOracle(p); ro.new_clip()
    for f, d in zip(frames, depths):
        got = R.render_frame(T(f), T(d), p).cpu().numpy()
        exp = ro.render(f, d, 0)
        assert np.array_equal(got, exp), u8_diff_stats(got, exp)
    S = R.debug_planes(p.warp_h, p.warp_w, p.eye_h, p.eye_w)["S"]
    assert float(S.abs().max()) * (p.warp_w - 1) / 2 > 50.0   # the clip does exercise shifts that reach three to four chunks


def test_singleton_state_leak_and_export_import(R, oracle):
    g = load_golden("render_loop.npz")
    sh, sw, n, kw = golden_json(g, "cases_json")["half_sbs_cli"]
    R.reset_state()
    _run_loop_hip(R, sh, sw, n, kw)
    st = R.export_state()
    got2, _, _ = _run_loop_hip(R, sh, sw, n, kw)  # second render, singletons NOT reset
    assert_parity("half_sbs_cli_second", *u8_diff_stats(got2, g["half_sbs_cli__second_render_frames"]))
    # state round-trip: importing the exported state reproduces the second render bit-for-bit
    R.reset_state()
    R.import_state(st)
    got3, _, _ = _run_loop_hip(R, sh, sw, n, kw)
    assert np.array_equal(got2, got3)


def test_render_frame_f32_depth_and_collapse(R, oracle):
    """precomputed float32 depth (BASELINE configs 1/3) + a constant depth frame (DepthPercentileEMA collapse guard)."""
    sh, sw = 108, 192
    kw = dict(output_format="Half-SBS", output_height=108, fg_shift=10.0, mg_shift=-2.5, bg_shift=-5.0, sharpness_factor=0.15,
              dof_strength=2.0, feather_strength=10.0, blur_ksize=9, use_subject_tracking=True, use_floating_window=True)
    p = render_kwargs_to_params(sw, sh, **kw)
    frames, depths = synth.synth_clip(4, sh, sw)
    depths[1] = np.full((sh, sw), 0.4, np.float32)
    R.reset_state(); R.new_clip()
    ro = oracle.RenderOracle(p); ro.new_clip()
    for i, (f, d) in enumerate(zip(frames, depths)):
        got = R.render_frame(T(f), T(d), p).cpu().numpy()
        exp = ro.render(f, d, 0)
        assert R.last_scalars().as_dict() == ro.last.as_dict(), i
        assert np.array_equal(got, exp), (i, u8_diff_stats(got, exp))


@pytest.mark.parametrize("ih,iw,H,W,edge", [(108, 192, 108, 192, True), (54, 96, 108, 192, True), (135, 240, 270, 480, False), (270, 480, 540, 960, True),
                                            (67, 101, 133, 203, True)])
def test_pixel_shift_without_feathering_keeps_the_shift_plane(R, oracle, ih, iw, H, W, edge):
    """Round 6: without feathering W1 computes the shift values of its own tile (k_shift's arithmetic folded into k_warp_fused<.., SHIFT = true>) and there is no
    k_shift launch.  pixel_shift_cuda's callers can still ask for the shift map: the folded kernel writes it, and it has to be the plane k_shift would have written
    (= the oracle's, which is pinned against the live reference's own shift map in pixel_shift_cases.npz) -- with and without the edge mask, with eyes at warp
    resolution and at half of it, on a size that is no multiple of the tile."""
    bgr, d = synth.synth_frame(3, ih, iw)
    ft = oracle.frame_to_tensor(bgr)
    for fs in (0.0, -2.0):
        p = ShiftParams.defaults(8.0, -2.0, -5.0, feather_strength=fs, blur_ksize=1, enable_edge_masking=edge, max_pixel_shift_percent=0.05)
        st = State()
        o = oracle.pixel_shift(ft, d[None], W, H, p, st, want_shift=True)
        R.reset_state()
        L, Rr, S = R.pixel_shift(T(ft), T(d[None]), W, H, p, want_shift=True)
        assert np.array_equal(S.cpu().numpy(), o["shift"]), (fs, float(np.abs(S.cpu().numpy() - o["shift"]).max()))
        assert np.array_equal(L.cpu().numpy(), o["left"]) and np.array_equal(Rr.cpu().numpy(), o["right"]), fs


def test_pixel_shift_without_feathering_in_the_scalar_tail_mode_goes_through_k_shift(R, oracle):
    """The folded kernel does not instantiate ATen's scalar-tail arithmetic (glibc expf / libm pow on the last elements of each thread's chunk): in the N-thread ATen
    mode on a plane that HAS tails, vd_warp_fold_ok refuses and k_shift<true> + the unfolded W1 run.  Same bytes as the oracle in both modes (the few tail elements of this plane happen to round
    alike in both arithmetics; tests/test_hip_widen.py's aten_any_size fixtures are where the mode shows)."""
    ih, iw, H, W = 67, 101, 133, 203
    bgr, d = synth.synth_frame(5, ih, iw)
    ft = oracle.frame_to_tensor(bgr)
    planes = {}
    for threads in (0, 3):
        p = ShiftParams.defaults(8.0, -2.0, -5.0, feather_strength=0.0, blur_ksize=1, max_pixel_shift_percent=0.05, aten_threads=threads)
        o = oracle.pixel_shift(ft, d[None], W, H, p, State(), want_shift=True)
        R.reset_state()
        L, Rr, S = R.pixel_shift(T(ft), T(d[None]), W, H, p, want_shift=True)
        assert np.array_equal(S.cpu().numpy(), o["shift"]), threads
        assert np.array_equal(L.cpu().numpy(), o["left"]) and np.array_equal(Rr.cpu().numpy(), o["right"]), threads
        planes[threads] = o["shift"]


def test_w1_row_tables_survive_more_geometries_than_the_cache_holds(R, oracle):
    """Round 6: W1's per-row parameters live in a per-geometry device table (k_wf_rowtab), eight geometries per process; the ninth replaces the oldest.  Eleven frame
    heights in a row, then the first one again (its table was evicted and is rebuilt): every eye pair equals the oracle's."""
    sizes = [(40 + 6 * i, 96) for i in range(11)] + [(40, 96)]
    for (H, W) in sizes:
        bgr, d = synth.synth_frame(H, H // 2, W // 2)
        ft = oracle.frame_to_tensor(bgr)
        p = ShiftParams.defaults(6.0, -2.0, -4.0, feather_strength=(0.0 if H % 4 else 10.0), blur_ksize=5)
        o = oracle.pixel_shift(ft, d[None], W, H, p, State())
        R.reset_state()
        L, Rr = R.pixel_shift(T(ft), T(d[None]), W, H, p)[:2]
        assert np.array_equal(L.cpu().numpy(), o["left"]) and np.array_equal(Rr.cpu().numpy(), o["right"]), (H, W)


KW_GUI = dict(output_format="Full-SBS", fg_shift=4.5, mg_shift=-1.5, bg_shift=-6.0, sharpness_factor=0.2, dof_strength=2.0, feather_strength=0.0,
              blur_ksize=1, use_subject_tracking=True, use_floating_window=True, zero_parallax_strength=0.01)   # VisionDepth3D.py:1405-1453


@pytest.mark.parametrize("fmt,size,fs,k", [("Full-SBS", (108, 192), 0.0, 1), ("Full-SBS", (270, 480), 0.0, 9), ("Half-SBS", (216, 384), 0.0, 1),
                                           ("Passive Interlaced", (108, 192), -3.0, 5), ("Red-Cyan Anaglyph", (144, 256), 0.0, 3),
                                           ("Half-SBS", (1080, 1920), 0.0, 1), ("Full-SBS", (1080, 1920), 0.0, 1)])
def test_feather_strength_zero_takes_the_exact_no_feather_warp(R, oracle, fmt, size, fs, k):
    """Round 5: with feather_strength <= 0 feather_shift_edges (core/render_3d.py:328-374) is an exact no-op, so the library warps without the
    mask kernel, the window sums and the blend (vd3d_api.hip::warp_stage_params) -- the GUI's own default configuration (VisionDepth3D.py:1405-1453).
    The oracle still goes the long way (and equals the live reference there: tests/test_oracle_vs_live_reference.py::
    test_render_loop_feather_strength_zero_exact); vd3d_debug_tune(4, 1) makes the library go the long way too.  All three: identical bytes, on
    the per-frame entry point and on the batched step path."""
    from visiondepth3d_amd import _lib
    sh, sw = size
    kw = dict(KW_GUI, output_format=fmt, output_height=sh, feather_strength=fs, blur_ksize=k)
    if fmt == "Full-SBS":
        kw.update(preserve_original_aspect=True, original_video_width=sw, original_video_height=sh)
    p = render_kwargs_to_params(sw, sh, **kw)
    n = 3 if sh >= 1080 else 5
    frames, depths = synth.synth_clip(n, sh, sw)
    ro = oracle.RenderOracle(p); ro.new_clip()
    exp = [ro.render(f, synth.depth_to_u8_bgr(d), 1) for f, d in zip(frames, depths)]
    outs = {}
    try:
        for long_way in (0, 1):
            _lib.check(_lib.lib().vd3d_debug_tune(4, long_way))
            R.reset_state(); R.new_clip()
            outs[long_way] = [R.render_frame(T(f), T(synth.depth_to_u8_bgr(d)), p).cpu().numpy() for f, d in zip(frames, depths)]
    finally:
        _lib.check(_lib.lib().vd3d_debug_tune(4, 0))
    for i in range(n):
        assert np.array_equal(outs[0][i], exp[i]), (i, u8_diff_stats(outs[0][i], exp[i]))
        assert np.array_equal(outs[1][i], exp[i]), (i, "long way", u8_diff_stats(outs[1][i], exp[i]))
    # the batched step path (render_pairs -> ChunkSharder at world 1 -> vd3d_shard_pixels) takes the same decision
    from visiondepth3d_amd.render_3d import render_pairs
    R.reset_state()
    kw_pairs = {k_: v for k_, v in kw.items()}
    got = list(render_pairs(zip(frames, [synth.depth_to_u8_bgr(d) for d in depths]), renderer=R, skip_first=False, batch=2, **kw_pairs))
    assert len(got) == n
    for i in range(n):
        assert np.array_equal(got[i], exp[i]), (i, "step path", u8_diff_stats(got[i], exp[i]))


@pytest.mark.parametrize("fmt,size,threads", [("Half-SBS", (108, 192), 1), ("Half-SBS", (108, 192), 6), ("Half-SBS", (1080, 1920), 4), ("Half-SBS", (1080, 1920), 64),
                                              ("Full-SBS", (540, 960), 3), ("Red-Cyan Anaglyph", (1080, 1920), 16), ("Half-SBS", (2160, 3840), 8),
                                              ("Half-SBS", (2160, 3840), 128), ("Half-SBS", (1080, 1920), 300)])
def test_aten_sum_order_of_the_two_torch_means(R, oracle, fmt, size, threads):
    """Round 5: vd3d_render_params::aten_sum_threads = N reproduces the float32 `torch.mean` of compute_dynamic_parallax_scale (:418) and compute_motion_metric
    (:928) as torch computes them with N intra-op threads -- ATen's cascade sum over the thread partition (vd3d_atensum.hip: one wave per crop row, one
    workgroup per thread share of the difference plane, level-0 blocks built in parallel and folded in order) -- bit for bit like the oracle's restatement,
    which tests/test_aten_restatements.py pins against torch itself.  Per-frame entry point and the batched step path (measure / replay: the float sums
    travel in the exchanged record), frames, scalars and final tracker state."""
    from visiondepth3d_amd.render_3d import render_pairs
    sh, sw = size
    kw = dict(KW_GUI, output_format=fmt, output_height=sh, feather_strength=6.0, blur_ksize=5, aten_sum_threads=threads)
    if fmt == "Full-SBS":
        kw.update(preserve_original_aspect=True, original_video_width=sw, original_video_height=sh)
    p = render_kwargs_to_params(sw, sh, **kw)
    assert p.aten_sum_threads == threads
    n = 3 if sh >= 1080 else 5
    frames, depths = synth.synth_clip(n, sh, sw)
    gray = [synth.depth_to_u8_bgr(d) for d in depths]
    ro = oracle.RenderOracle(p); ro.new_clip()
    exp, exp_sc = [], []
    for f, g in zip(frames, gray):
        exp.append(ro.render(f, g, 1)); exp_sc.append(ro.last.as_dict())
    R.reset_state(); R.new_clip()
    for i, (f, g) in enumerate(zip(frames, gray)):
        got = R.render_frame(T(f), T(g), p).cpu().numpy()
        a, b = R.last_scalars().as_dict(), exp_sc[i]
        assert a == b, (i, {k_: (a[k_], b[k_]) for k_ in a if a[k_] != b[k_]})
        assert np.array_equal(got, exp[i]), (i, u8_diff_stats(got, exp[i]))
    st_seq = R.export_state().as_dict()
    assert st_seq == ro.state.as_dict()
    R.reset_state()
    got = list(render_pairs(zip(frames, gray), renderer=R, skip_first=False, batch=2, **kw))
    for i in range(n):
        assert np.array_equal(got[i], exp[i]), (i, "step path", u8_diff_stats(got[i], exp[i]))
    assert R.export_state().as_dict() == st_seq
    # the order-free default (aten_sum_threads = 0) is a different -- correctly rounded -- mean: same frames on almost every clip, never the subject here
    p0 = render_kwargs_to_params(sw, sh, **dict(kw, aten_sum_threads=0))
    assert p0.aten_sum_threads == 0


@pytest.mark.parametrize("fmt,size,oh,threads", [("Half-SBS", (72, 128), 72, 4), ("Half-SBS", (90, 150), 60, 8), ("Full-SBS", (54, 70), 54, 2), ("Red-Cyan Anaglyph", (75, 133), 50, 1),
                                                  ("Half-SBS", (270, 480), 270, 1), ("Passive Interlaced", (101, 203), 101, 3)])
def test_aten_mode_at_sizes_with_scalar_tails_and_small_eyes(R, oracle, fmt, size, oh, threads):
    """Round 5, the N-thread ATen mode beyond the two means: planes whose element count is not a multiple of 32 send the last elements of every thread's chunk
    through libm in torch.pow / torch.sigmoid (k_chain_shape<true>, k_shift<true>: fp64 pow and glibc's expf), and eyes / warp planes with H + W <= 128 -- or the
    3-channel frame whenever the reference ran ONE torch thread, at any size -- are resized by ATen's premultiplied-weight bilinear kernel (ingest and chain
    kernels by flag; W1 and the finishing kernels get the plane resized ahead of them and run with identity geometry).  The oracle's restatement of both is pinned
    against the live reference at arbitrary sizes (tests/test_oracle_vs_live_reference.py::test_render_loop_any_size_exact_in_aten_mode); here device == oracle:
    frames, scalars, state -- per-frame entry point, the batched step path, and the stand-alone finishing entry."""
    from visiondepth3d_amd.render_3d import render_pairs
    sh, sw = size
    kw = dict(KW_GUI, output_format=fmt, output_height=oh, feather_strength=7.0, blur_ksize=5, dof_strength=2.0, aten_sum_threads=threads)
    if fmt == "Full-SBS":
        kw.update(preserve_original_aspect=True, original_video_width=sw, original_video_height=sh)
    p = render_kwargs_to_params(sw, sh, **kw)
    n = 4
    frames, depths = synth.synth_clip(n, sh, sw)
    gray = [synth.depth_to_u8_bgr(d) for d in depths]
    ro = oracle.RenderOracle(p); ro.new_clip()
    exp, exp_sc = [], []
    for f, g in zip(frames, gray):
        exp.append(ro.render(f, g, 1)); exp_sc.append(ro.last.as_dict())
    R.reset_state(); R.new_clip()
    for i, (f, g) in enumerate(zip(frames, gray)):
        got = R.render_frame(T(f), T(g), p).cpu().numpy()
        a, b = R.last_scalars().as_dict(), exp_sc[i]
        assert a == b, (i, {k_: (a[k_], b[k_]) for k_ in a if a[k_] != b[k_]})
        assert np.array_equal(got, exp[i]), (i, u8_diff_stats(got, exp[i]))
    assert R.export_state().as_dict() == ro.state.as_dict()
    R.reset_state()
    got = list(render_pairs(zip(frames, gray), renderer=R, skip_first=False, batch=2, **kw))
    for i in range(n):
        assert np.array_equal(got[i], exp[i]), (i, "step path", u8_diff_stats(got[i], exp[i]))
    # the default mode is a different (thread-independent) arithmetic on exactly these corners: it must NOT be what the ATen mode computes everywhere
    p0 = render_kwargs_to_params(sw, sh, **dict(kw, aten_sum_threads=0))
    ro0 = oracle.RenderOracle(p0); ro0.new_clip()
    R.reset_state(); R.new_clip()
    for f, g in zip(frames, gray):
        assert np.array_equal(R.render_frame(T(f), T(g), p0).cpu().numpy(), ro0.render(f, g, 1))
    # stand-alone finishing entry: depth_for_dof resized by the kernel ATen picks
    rng = np.random.default_rng(sh)
    Lh = rng.integers(0, 256, (p.warp_h, p.warp_w, 3), dtype=np.uint8); Rh = rng.integers(0, 256, (p.warp_h, p.warp_w, 3), dtype=np.uint8)
    dn = rng.random((p.eye_h, p.eye_w), dtype=np.float32)
    got_f = R.finish_frame(T(Lh), T(Rh), T(dn), p, 0.4).cpu().numpy()
    assert np.array_equal(got_f, oracle.finish_frame(Lh, Rh, dn, p, 0.4, 0, 0))


def test_other_entry_points_between_the_frames_of_a_step_path_clip(R, oracle):
    """`format_3d_output(..., "VR")` and `pixel_shift_cuda` at ANOTHER size re-size the context's shared warp-resolution planes; a module-level
    call of either on the renderer that is in the middle of a batched `render_pairs` clip (frames arrive up to two steps late, so a consumer does exactly
    that between yields) used to leave the next pixel pass writing H x W planes into the smaller buffers.  Now format_3d_output has a scratch of its own and the
    pixel pass re-establishes its planes: the clip's frames stay the oracle's, and the interleaved calls return their own right answers."""
    from visiondepth3d_amd.render_3d import render_pairs
    sh, sw = 270, 480
    kw = dict(KW_GUI, output_format="Full-SBS", output_height=sh, feather_strength=8.0, blur_ksize=5, preserve_original_aspect=True,
              original_video_width=sw, original_video_height=sh)
    p = render_kwargs_to_params(sw, sh, **kw)
    n = 9
    frames, depths = synth.synth_clip(n, sh, sw)
    gray = [synth.depth_to_u8_bgr(d) for d in depths]
    ro = oracle.RenderOracle(p); ro.new_clip()
    exp = [ro.render(f, g, 1) for f, g in zip(frames, gray)]
    rng = np.random.default_rng(3)
    eyeL = rng.integers(0, 256, (90, 160, 3), dtype=np.uint8)
    eyeR = rng.integers(0, 256, (90, 160, 3), dtype=np.uint8)
    small = synth.synth_frame(3, 54, 96)
    # (no tracker in the interleaved call: pixel_shift_cuda's FloatingWindowTracker is shared state in the reference too, and this test is about memory)
    sp = ShiftParams.defaults(6.0, -1.5, -4.0, blur_ksize=5, feather_strength=8.0, use_subject_tracking=False, enable_floating_window=False)
    ft = torch.from_numpy(oracle.frame_to_tensor(small[0])).cuda()
    dt = torch.from_numpy(small[1])[None].cuda()
    R.reset_state()
    got = []
    for i, fr in enumerate(render_pairs(zip(frames, gray), renderer=R, skip_first=False, batch=2, **kw)):
        got.append(fr)
        vr = R.format_3d_output(T(eyeL), T(eyeR), "VR").cpu().numpy()            # 1440 x 1600 eyes: a different plane size
        assert np.array_equal(vr, oracle.format_output(eyeL, eyeR, 2)), i
        if i % 2 == 0:
            L_, R_ = R.pixel_shift(ft, dt, 96, 54, sp)                           # ... and a smaller one, through the shared planes themselves
            assert L_.shape == (54, 96, 3) and R_.shape == (54, 96, 3)
    assert len(got) == n
    for i in range(n):
        assert np.array_equal(got[i], exp[i]), (i, u8_diff_stats(got[i], exp[i]))


KW_CLI = dict(output_format="Half-SBS", fg_shift=10.0, mg_shift=-2.5, bg_shift=-5.0, sharpness_factor=0.15, dof_strength=2.0,
              feather_strength=10.0, blur_ksize=9, use_subject_tracking=True, use_floating_window=True)


def _seq_render(R, p, frames, depths, blank=()):
    R.reset_state(); R.new_clip()
    out, sc = [], []
    for t, (f, d) in enumerate(zip(frames, depths)):
        out.append(R.render_frame(f, d, p, blank=t in blank).cpu().numpy())
        sc.append(R.last_scalars().as_dict())
    return out, sc, R.export_state().as_dict(), R.tdf_plane_export(p).cpu().numpy()


def _emulated_ranks(p, G, B):
    from shard_emul import Emu
    from visiondepth3d_amd.render_3d import Renderer
    from visiondepth3d_amd.sharded import ChunkSharder, HipChunkBackend
    ranks = [Renderer(0) for _ in range(G)]
    shd = []
    for g_, rr in enumerate(ranks):
        rr.reset_state(); rr.new_clip()
        shd.append(ChunkSharder(HipChunkBackend(rr, p), g_, G, B))
    return ranks, Emu(shd)


@pytest.mark.parametrize("G,B", [(2, 3), (3, 2), (4, 1)])
def test_chunk_sharding_equals_sequential(R, oracle, G, B):
    """ChunkSharder (contiguous chunks, vd3d_shard2_* + vd3d_tdf_plane_export / import): G contexts play the ranks, the plane
    hand-off and the two all-gathers are done by hand (tests/shard_emul.py).  Muxed frames, the final tracker state and the final
    plane state of EVERY rank must equal the sequential render bit for bit.  Three steps (state carried across steps and chunk
    boundaries, first step starts a clip), incl. a frame whose depth collapses (DepthPercentileEMA guard path)."""
    sh, sw = 108, 192
    p = render_kwargs_to_params(sw, sh, output_height=108, **KW_CLI)
    n = 3 * B * G
    frames, depths = synth.synth_clip(n, sh, sw)
    gray = [synth.depth_to_u8_bgr(d)[..., 0].copy() for d in depths]
    gray[2 * B * G - 1][:] = 77
    ft, dt = [T(f) for f in frames], [T(d) for d in gray]
    seq, _, st_seq, plane_seq = _seq_render(R, p, ft, dt)
    ranks, emu = _emulated_ranks(p, G, B)
    got = emu.run_clip(ft, dt, B)
    assert sorted(got) == list(range(n))
    for t in range(n):
        assert np.array_equal(got[t].cpu().numpy(), seq[t]), (t, u8_diff_stats(got[t].cpu().numpy(), seq[t]))
    for rr in ranks:
        st = rr.export_state().as_dict()
        assert st == st_seq, {k: (st[k], st_seq[k]) for k in st if st[k] != st_seq[k]}
        assert np.array_equal(rr.tdf_plane_export(p).cpu().numpy(), plane_seq)
        rr.close()


def test_chunk_sharding_partial_last_step_and_blank_frames(R, oracle):
    """A clip whose length is not a multiple of world * B (the last step is partial: one rank has a short chunk, one has none and
    only forwards the plane) with skip_blank_frames hits in both steps: blank own frames take the blank pixel pass, the replay skips
    the ipd scaling / focal tracker / FloatingWindowTracker for them on EVERY rank (core/render_3d.py:1278-1281,1334-1337)."""
    sh, sw, G, B = 108, 192, 3, 2
    p = render_kwargs_to_params(sw, sh, output_height=108, skip_blank_frames=True, ipd_factor=1.15, **KW_CLI)
    n = G * B + 3                    # one full step + a partial one: rank 0 two frames, rank 1 one, rank 2 none
    frames, depths = synth.synth_clip(n, sh, sw)
    gray = [synth.depth_to_u8_bgr(d)[..., 0].copy() for d in depths]
    blank = {1, 4, 7}
    for t in blank:
        frames[t][:] = (frames[t] // 16)        # dark frames, like the ones ffmpeg's blackdetect reports
    ft, dt = [T(f) for f in frames], [T(d) for d in gray]
    seq, _, st_seq, plane_seq = _seq_render(R, p, ft, dt, blank)
    ranks, emu = _emulated_ranks(p, G, B)
    got = emu.run_clip(ft, dt, B, blank_frames=blank)
    assert sorted(got) == list(range(n))
    for t in range(n):
        assert np.array_equal(got[t].cpu().numpy(), seq[t]), t
    for rr in ranks:
        assert rr.export_state().as_dict() == st_seq
        assert np.array_equal(rr.tdf_plane_export(p).cpu().numpy(), plane_seq)
        rr.close()


def test_chunk_sharding_world1_and_continuation(R, oracle):
    """world = 1 degenerates to the sequential render; a sequential frame rendered AFTER sharded steps continues exactly."""
    from visiondepth3d_amd.sharded import ChunkSharder, HipChunkBackend
    sh, sw = 108, 192
    p = render_kwargs_to_params(sw, sh, output_height=108, **KW_CLI)
    frames, depths = synth.synth_clip(9, sh, sw)
    gray = [synth.depth_to_u8_bgr(d)[..., 0].copy() for d in depths]
    ft, dt = [T(f) for f in frames], [T(d) for d in gray]
    seq, _, st_seq, _ = _seq_render(R, p, ft, dt)
    R.reset_state(); R.new_clip()
    one = ChunkSharder(HipChunkBackend(R, p), 0, 1, 4)
    outs = [o.cpu().numpy() for o in one.render_step(ft[:4], dt[:4], first_step=True)]
    outs += [o.cpu().numpy() for o in one.render_step(ft[4:8], dt[4:8])]
    outs.append(R.render_frame(ft[8], dt[8], p).cpu().numpy())
    assert all(np.array_equal(a, b) for a, b in zip(outs, seq))
    assert R.export_state().as_dict() == st_seq


@pytest.mark.parametrize("two_sets", [True, False])
def test_overlapped_pixel_passes_equal_sequential(R, oracle, two_sets):
    """vd3d_set_pixel_overlap: pixel passes on the context's second stream while the next step's measurement chain runs on the
    first.  Two alternating slot sets (real overlap) and ONE reused slot set (every chain call first waits for the pixel pass that
    still reads its slot) both reproduce the sequential render bit for bit; an unsharded frame afterwards continues exactly."""
    from visiondepth3d_amd.sharded import ChunkSharder, HipChunkBackend
    sh, sw, B, steps = 270, 480, 3, 4
    p = render_kwargs_to_params(sw, sh, output_height=sh, **KW_CLI)
    n = B * steps + 1
    frames, depths = synth.synth_clip(n, sh, sw)
    ft, dt = [T(f) for f in frames], [T(d) for d in depths]
    seq, _, st_seq, _ = _seq_render(R, p, ft, dt)
    R.reset_state(); R.new_clip()
    be = HipChunkBackend(R, p)
    first = ChunkSharder(be, 0, 1, B)
    sets = [first, ChunkSharder(be, 0, 1, B, slot_base=B, twin_of=first)] if two_sets else [first] * 2
    R.set_pixel_overlap(True)
    try:
        outs = []
        for i in range(steps):
            outs += sets[i % 2].render_step(ft[i * B:(i + 1) * B], dt[i * B:(i + 1) * B], first_step=(i == 0))
        last = R.render_frame(ft[-1], dt[-1], p)     # joins the outstanding pixel passes before it touches L / R / S
        R.sync()
        got = [o.cpu().numpy() for o in outs] + [last.cpu().numpy()]
    finally:
        R.set_pixel_overlap(False)
    assert len(got) == len(seq) and all(np.array_equal(a, b) for a, b in zip(got, seq))
    assert R.export_state().as_dict() == st_seq


def test_render_clip_overlapped_pixels(R, oracle):
    """ChunkSharder.render_clip(overlap_pixels=True): frames are yielded one step late, after a host wait on their pixel pass; whole
    clip incl. a partial last step and a blank frame == the sequential render, and the context is left in sequential mode.  A second
    clip on the same sharder starts from fresh per-clip state (render_clip calls new_clip itself)."""
    from visiondepth3d_amd.sharded import ChunkSharder, HipChunkBackend
    sh, sw, B, n = 270, 480, 3, 11
    p = render_kwargs_to_params(sw, sh, output_height=sh, skip_blank_frames=True, **KW_CLI)
    frames, depths = synth.synth_clip(n, sh, sw)
    ft, dt = [T(f) for f in frames], [T(d) for d in depths]
    seq, _, st_seq, _ = _seq_render(R, p, ft, dt, {5})
    R.reset_state()
    shr = ChunkSharder(HipChunkBackend(R, p), 0, 1, B)
    got = [(t, o.cpu().numpy()) for t, o in shr.render_clip(n, lambda t: ft[t], lambda t: dt[t], blank_frames={5}, overlap_pixels=True)]
    assert [t for t, _ in got] == list(range(n))
    assert all(np.array_equal(o, seq[t]) for t, o in got)
    assert R.export_state().as_dict() == st_seq
    # a second clip on the same sharder (sequential mode): fresh per-clip state, the never-reset singletons carry over (:500)
    again = [o.cpu().numpy() for _, o in shr.render_clip(n, lambda t: ft[t], lambda t: dt[t], blank_frames={5})]
    R.reset_state()
    for _rep in range(2):
        R.new_clip()
        seq2 = [R.render_frame(ft[t], dt[t], p, blank=(t == 5)).cpu().numpy() for t in range(n)]
    assert all(np.array_equal(a, b) for a, b in zip(again, seq2))


# ------------------------------------------------------------------------------------------ full-size
@pytest.mark.parametrize("hw", [(1080, 1920), (2160, 3840)])
def test_full_size_properties(R, oracle, hw):
    sh, sw = hw
    kw = dict(output_format="Half-SBS", output_height=sh, fg_shift=10.0, mg_shift=-2.5, bg_shift=-5.0, sharpness_factor=0.15,
              dof_strength=2.0, feather_strength=10.0, blur_ksize=9, use_subject_tracking=True, use_floating_window=True)
    p = render_kwargs_to_params(sw, sh, **kw)
    assert (p.warp_w, p.warp_h, p.eye_w, p.eye_h, p.out_w, p.out_h) == (sw, sh, sw // 2, sh // 2, sw, sh)
    f, d = synth.synth_frame(0, sh, sw)
    ft, dt = T(f), T(d)
    R.reset_state(); R.new_clip()
    a = R.render_frame(ft, dt, p).clone()
    pl = R.debug_planes(sh, sw, p.eye_h, p.eye_w)
    # determinism: same state + same input -> identical bytes (integer atomics / fixed-point sums only)
    R.reset_state(); R.new_clip()
    b = R.render_frame(ft, dt, p)
    assert torch.equal(a, b)
    # order statistics against torch's sort-based implementation at full size
    sc = R.last_scalars()
    dn = pl["dn"]
    dfilt = torch.from_numpy(oracle.interp_bilinear(d[None], p.eye_h, p.eye_w)).cuda().clamp(0, 1)
    assert sc.q_lo == float(torch.quantile(dfilt.flatten(), float(np.float32(0.02))))
    assert sc.q_hi == float(torch.quantile(dfilt.flatten(), float(np.float32(0.98))))
    assert float(dn.min()) >= 0.0 and float(dn.max()) <= 1.0
    # zero layer shifts, no tracking, no convergence -> both eyes sample the same grid: L == R exactly, SBS halves equal
    kw0 = dict(kw, fg_shift=0.0, mg_shift=0.0, bg_shift=0.0, use_subject_tracking=False, use_floating_window=False)
    p0 = render_kwargs_to_params(sw, sh, **kw0)
    R.reset_state(); R.new_clip()
    z = R.render_frame(ft, dt, p0)
    assert torch.equal(z[:, : sw // 2], z[:, sw // 2:])
    # one full-size bit-exact frame against the oracle with the configuration's OWN input type, float32 precomputed depth (BASELINE
    # configs[2] at 3840x2160: ~6 s of oracle time)
    ro = oracle.RenderOracle(p); ro.new_clip()
    exp = ro.render(f, d, 0)
    assert np.array_equal(a.cpu().numpy(), exp), u8_diff_stats(a.cpu().numpy(), exp)


# ------------------------------------------------------------------------------------------ B2 attribution on the GPU
def test_b2_attribution_hip_finish_stage_vs_reference_frames(R):
    """vd3d_finish_frame fed with the REFERENCE's own eyes (tests/golden/attrib.npz) against the reference's own muxed frames: with
    dof_dense_conv every sample of all 11 frames is EXACT; with the separable default the committed, measured deviation holds
    (tests/test_oracle_vs_golden.py::test_b2_attribution_finish_stage_separable_is_measured)."""
    from test_oracle_vs_golden import attrib_stats

    def finish(L, R_, dn, p, focal, bar_w, bar_side):
        return R.finish_frame(T(L), T(R_), T(dn), p, focal, bar_w, bar_side).cpu().numpy()
    for name, (mx, ne, n1, tot) in attrib_stats(finish, True).items():
        assert (mx, ne) == (0, 0), (name, mx, ne, tot)
    st = attrib_stats(finish, False)
    assert st["half_sbs_cli"][0] <= 4 and st["half_sbs_cli"][1] <= 2000 and st["half_sbs_cli"][2] <= 800, st
    assert st["half_sbs_graded"][1] <= 20 and st["full_sbs_preserve"][1] <= 60, st


def test_configs0_real_size_1080p_hip_vs_reference_fixture_and_oracle(R, oracle):
    """BASELINE configs[0] at real size through the C ABI: vs the reference's frames (real1080.npz; same bars as the oracle test) in
    both DOF modes, and bit-exact vs the oracle in the dense mode (the separable mode is covered at 1080p by test_full_size_properties)."""
    from test_oracle_vs_golden import real1080_stats
    kept = {}

    def render(p, frames, dbgr):
        R.reset_state(); R.new_clip()
        outs = [R.render_frame(T(f), T(d), p).cpu().numpy() for f, d in zip(frames, dbgr)]
        kept[int(p.dof_dense_conv)] = (p, frames, dbgr, outs)
        return outs
    stats = real1080_stats(render)
    for s in stats(True):
        assert_parity("real_half_sbs", s["max"], s["ne"] / s["n"], s["n1"] / s["n"])
        assert s["rowsum"] <= 8, s
    for s in stats(False):
        assert s["max"] <= 4 and s["ne"] <= 3000 and s["n1"] <= 1300, s
    p, frames, dbgr, outs = kept[1]
    ro = oracle.RenderOracle(p)
    ro.new_clip()
    assert np.array_equal(outs[0], ro.render(frames[0], dbgr[0], 1))


def test_configs2_real_size_4k_hip_vs_reference_fixture(R):
    """BASELINE configs[2] at real size through the C ABI: both rendered frames of the 3840x2160 clip equal the live reference's frames
    (tests/golden/real4k.npz: SHA-256 of the whole frame, bands, decimated copy, row / column sums)."""
    from test_oracle_vs_golden import real4k_check

    def render(p, frames, dbgr):
        R.reset_state(); R.new_clip()
        return [R.render_frame(T(f), T(d), p).cpu().numpy() for f, d in zip(frames, dbgr)]
    real4k_check(render, 2)


def test_real_size_4k_other_formats_hip_vs_reference_fixture(R):
    """Full-SBS (preserve), Passive Interlaced, Red-Cyan Anaglyph and VR at 3840x2160 through the C ABI: both rendered frames of each equal the
    live reference's (tests/golden/real4k_formats.npz: SHA-256 of the whole frame, row / column sums, bands)."""
    from test_oracle_vs_golden import real4k_formats_check

    def render(p, frames, dbgr):
        R.reset_state(); R.new_clip()
        return [R.render_frame(T(f), T(d), p).cpu().numpy() for f, d in zip(frames, dbgr)]
    real4k_formats_check(render, 2)
    # DOF strengths beyond the fused finishing kernel's 9 taps at real size: 3.0 (13 taps, Half-SBS) and the slider's maximum 5.0 (21 taps,
    # anaglyph) -- k_dof_grade4's instantiations and k_sharp_fit against the reference's own 4K frames (tests/golden/real4k_dof.npz)
    real4k_formats_check(render, 2, fixture="real4k_dof.npz")


def test_real_size_1080p_random_configurations_hip_vs_reference_fixture(R):
    """Twelve random configurations of the render loop at 1920x1080 through the C ABI: every frame equals the live reference's
    (tests/golden/real1080_random.npz: SHA-256 of the whole frame, row sums)."""
    from test_oracle_vs_golden import real1080_random_check

    def render(p, frames, dbgr):
        R.reset_state(); R.new_clip()
        return [R.render_frame(T(f), T(d), p).cpu().numpy() for f, d in zip(frames, dbgr)]
    real1080_random_check(render)
    # five more of the same generator at 3840x2160 (tests/golden/real4k_random.npz): layer shifts up to 127 pixels, blur sizes up to 13, a 15-tap DOF,
    # VR, masking / feathering off -- the fused warp's chunk rule and the unfused fallbacks at real size against the reference's own 4K frames
    real1080_random_check(render, fixture="real4k_random.npz", sh=2160, sw=3840)


def test_gui_default_configuration_hip_vs_reference_fixture(R):
    """The GUI's own defaults (feather_strength 0.0: the exact no-feather warp; Full-SBS at 1920x1080 and -- through the 2 x 2 fit of the fused finishing
    kernel's vector epilogue -- at 3840x2160; Half-SBS; feather 0 behind a 9 x 9 window) through the C ABI: every frame equals the live reference's
    (tests/golden/gui_defaults.npz: SHA-256 of the whole frame, row sums), per-frame entry point and batched step path."""
    from test_oracle_vs_golden import gui_defaults_check
    from visiondepth3d_amd.sharded import ChunkSharder, HipChunkBackend

    def render(p, frames, dbgr):
        R.reset_state(); R.new_clip()
        return [R.render_frame(T(f), T(d), p).cpu().numpy() for f, d in zip(frames, dbgr)]
    gui_defaults_check(render)

    def render_steps(p, frames, dbgr):
        R.reset_state(); R.new_clip()
        sh = ChunkSharder(HipChunkBackend(R, p), 0, 1, len(frames))
        outs = sh.render_step([T(f) for f in frames], [T(d) for d in dbgr], first_step=True)
        return [o.cpu().numpy() for o in outs]
    gui_defaults_check(render_steps, names=("gui_4k_full", "gui_1080_half"))


def test_aten_mode_any_size_hip_vs_reference_fixture(R):
    """Round 5: ten frame sizes no size rule covers (odd sizes, cropped sources, thumbnails), each rendered by the live reference with a fixed torch thread count
    (tests/golden/aten_any_size.npz), through the C ABI in the N-thread ATen mode -- libm on ATen's scalar tails, ATen's premultiplied-weight bilinear kernel for the
    small planes and for the frame of a one-thread reference, the cascade sums: every frame equals the reference's, per-frame entry point and batched step path."""
    from test_oracle_vs_golden import aten_any_size_check
    from visiondepth3d_amd.sharded import ChunkSharder, HipChunkBackend

    def render(p, frames, dbgr):
        R.reset_state(); R.new_clip()
        return [R.render_frame(T(f), T(d), p).cpu().numpy() for f, d in zip(frames, dbgr)]
    assert aten_any_size_check(render) == 10

    def render_steps(p, frames, dbgr):
        R.reset_state(); R.new_clip()
        sh = ChunkSharder(HipChunkBackend(R, p), 0, 1, len(frames))
        outs = sh.render_step([T(f) for f in frames], [T(d) for d in dbgr], first_step=True)
        return [o.cpu().numpy() for o in outs]
    aten_any_size_check(render_steps)


def test_real_size_1080p_letterbox_auto_crop_hip_vs_reference_fixture(R):
    """Black-bar auto crop at 1920x1080 through the C ABI (K0 k_autocrop per frame, crop, re-fit): four letterboxed clips, every frame equals
    the live reference's (tests/golden/real1080_letterbox.npz: SHA-256 of the whole frame, row sums)."""
    from test_oracle_vs_golden import real1080_letterbox_check

    def render(p, frames, dbgr):
        R.reset_state(); R.new_clip()
        return [R.render_frame(T(f), T(d), p).cpu().numpy() for f, d in zip(frames, dbgr)]
    real1080_letterbox_check(render)


def test_real_size_other_formats_hip_vs_reference_fixture_and_oracle(R, oracle):
    """Full-SBS (preserve), Passive Interlaced and Red-Cyan Anaglyph at 1920x1080 through the C ABI: against the live reference's
    frames (real1080_formats.npz) under the measured per-format ceilings of conftest.PARITY_BARS, and the first frame of every
    format bit-exact against the oracle."""
    from test_oracle_vs_golden import real_format_stats
    first = {}

    def render(name, p, frames, dbgr):
        R.reset_state(); R.new_clip()
        outs = [R.render_frame(T(f), T(d), p).cpu().numpy() for f, d in zip(frames, dbgr)]
        first[name] = (p, frames[0], dbgr[0], outs[0])
        return outs
    for name, per_frame in real_format_stats(render).items():
        for mx, fr, f1, rs in per_frame:
            assert_parity("real_" + name, mx, fr, f1)
    for name, (p, f, d, out) in first.items():
        ro = oracle.RenderOracle(p)
        ro.new_clip()
        exp = ro.render(f, d, 1)
        assert np.array_equal(out, exp), (name, u8_diff_stats(out, exp))


def test_render_clip_with_a_private_stream_renderer(oracle):
    """render_pairs / render_clip with Renderer(private_stream=True) must order its device-to-host copies behind the
    renderer's own stream (the copy runs on torch's current stream).  A long clip at 1080p keeps the private stream busy while the
    host races ahead; every frame must still equal the oracle's."""
    from visiondepth3d_amd.render_3d import Renderer, render_clip
    sh, sw, n = 540, 960, 6
    frames, depths = synth.synth_clip(n, sh, sw)
    kw = dict(output_format="Half-SBS", output_height=sh, fg_shift=10.0, mg_shift=-2.5, bg_shift=-5.0, sharpness_factor=0.15,
              dof_strength=2.0, feather_strength=10.0, blur_ksize=9, use_subject_tracking=True, use_floating_window=True)
    r = Renderer(0, private_stream=True)
    try:
        got = list(render_clip(frames, depths, renderer=r, **kw))
        r.reset_state()    # the second render starts from fresh trackers too (they persist across renders like the reference's singletons)
        dev = list(render_clip([T(f) for f in frames], [T(d) for d in depths], renderer=r, keep_on_device=True, **kw))
    finally:
        torch.cuda.synchronize()
        r.close()
    p = render_kwargs_to_params(sw, sh, **kw)
    ro = oracle.RenderOracle(p)
    ro.new_clip()
    exp = [ro.render(f, d, 0) for f, d in list(zip(frames, depths))[1:]]
    assert len(got) == len(exp) == n - 1
    for i, (a, b, e) in enumerate(zip(got, dev, exp)):
        assert np.array_equal(a, e), (i, u8_diff_stats(a, e))
        assert np.array_equal(b.cpu().numpy(), e), i
