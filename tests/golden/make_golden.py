#!/usr/bin/env python3
"""Generate the golden vectors by running the REFERENCE itself (development container only).

    python tests/golden/make_golden.py            # writes tests/golden/*.npz

Inputs are procedural (visiondepth3d_amd.synth) or integer formulas, so the fixtures hold
expected outputs + the parameters that produced them.  The reference is imported from
/root/reference through tests/golden/ref_loader.py (third-party modules that are absent here are
restated in ref_stubs.py -- those parts are "parity unpinned").
"""
from __future__ import annotations

import hashlib
import io
import contextlib
import json
import os
import sys
import threading

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import ref_loader as rl  # noqa: E402
import ref_stubs  # noqa: E402
from cases import HEAL_CASES, heal_inputs  # noqa: E402
from visiondepth3d_amd import synth  # noqa: E402

r = rl.load()
torch.set_num_threads(8)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrs)
    print(f"wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB")


# ------------------------------------------------------------------------------------------
# 1. SURVEY Appendix A known answers (integer-defined input)
# ------------------------------------------------------------------------------------------
def kat_inputs(H=144, W=256):
    y, x = np.mgrid[0:H, 0:W]
    bgr = np.stack([((3 * x + 5 * y + k) % 256) for k in (0, 7, 14)], axis=2).astype(np.uint8)
    d = (((4 * x + 3 * y) % 256) / 255).astype(np.float32)[None]
    return bgr, d


def gen_kat():
    bgr, d = kat_inputs()
    ft = r.frame_to_tensor(bgr)
    dt = torch.from_numpy(d)
    out = {}
    dc = r.enhance_curvature(dt, 0.08).clamp(0, 1)
    s0 = r.estimate_subject_depth(dc)
    out["s0"] = np.float32(s0.item())
    out["q05"] = np.float32(torch.quantile(dc, 0.05).item())
    out["q95"] = np.float32(torch.quantile(dc, 0.95).item())
    D = r.shape_depth_for_pop(dc, s0)
    out["D"] = D.numpy()
    out["s1"] = np.float32(r.estimate_subject_depth(D).item())
    out["dyn_scale"] = np.float64(r.compute_dynamic_parallax_scale(dt, 0.90, 1.15))
    for tag, kw in (("trk", dict(use_subject_tracking=True, enable_floating_window=True)),
                    ("notrk", dict(use_subject_tracking=False, enable_floating_window=False))):
        rl.reset_state()
        L, R, S = r.pixel_shift_cuda(ft, dt, 256, 144, 10, -2.5, -5, blur_ksize=9, feather_strength=10, **kw)
        out[f"{tag}_L"], out[f"{tag}_R"], out[f"{tag}_S"] = L, R, S.numpy()
        out[f"{tag}_prev_offset"] = np.float64(r.floating_window_tracker.prev_offset)
    out["grade_sum"] = np.float64(r.apply_color_grade(ft, 1.35, 1.10, 0.04).double().sum().item())
    save("kat_appendix_a.npz", **out)


# ------------------------------------------------------------------------------------------
# 2. pixel_shift_cuda parameter sweep on a synthetic frame
# ------------------------------------------------------------------------------------------
SHIFT_CASES = [
    ("cli_defaults", (10.0, -2.5, -5.0), dict()),
    ("gui_defaults", (4.5, -1.5, -6.0), dict(blur_ksize=1, feather_strength=0.0, zero_parallax_strength=0.01)),
    ("no_tracking", (10.0, -2.5, -5.0), dict(use_subject_tracking=False, enable_floating_window=False)),
    ("track_no_float", (10.0, -2.5, -5.0), dict(enable_floating_window=False, zero_parallax_strength=0.01)),
    ("conv_dynamic", (10.0, -2.5, -5.0), dict(convergence_strength=0.02)),
    ("conv_static", (10.0, -2.5, -5.0), dict(convergence_strength=-0.03, enable_dynamic_convergence=False)),
    ("no_edge_mask", (10.0, -2.5, -5.0), dict(enable_edge_masking=False)),
    ("no_feather", (10.0, -2.5, -5.0), dict(enable_feathering=False)),
    ("pop_controls", (8.0, -2.0, -4.0), dict(blur_ksize=4, feather_strength=3.0, depth_pop_gamma=0.7, depth_pop_mid=0.45,
                                             depth_stretch_lo=0.1, depth_stretch_hi=0.9, fg_pop_multiplier=1.4,
                                             bg_push_multiplier=0.9, subject_lock_strength=0.8, parallax_balance=0.6,
                                             max_pixel_shift_percent=0.05)),
    ("big_shift_k15", (30.0, -8.0, -20.0), dict(blur_ksize=15, feather_strength=20.0, max_pixel_shift_percent=0.08)),
]


def gen_pixel_shift():
    out = {}
    meta = {}
    # identity-size case (preview path: frame and depth already at warp size) and an upsampling case (render path)
    for size_tag, (ih, iw, H, W) in (("id", (96, 160, 96, 160)), ("up", (54, 96, 108, 192))):
        bgr, d = synth.synth_frame(3, ih, iw)
        ft = r.frame_to_tensor(bgr)
        dt = torch.from_numpy(d)[None]
        for name, (fg, mg, bg), kw in SHIFT_CASES:
            if size_tag == "up" and name not in ("cli_defaults", "gui_defaults", "pop_controls"):
                continue
            rl.reset_state()
            L, R, S = r.pixel_shift_cuda(ft, dt, W, H, fg, mg, bg, **kw)
            key = f"{size_tag}__{name}"
            out[key + "__L"], out[key + "__R"] = L, R
            out[key + "__S"] = S.numpy().astype(np.float32)
            out[key + "__prev_offset"] = np.float64(r.floating_window_tracker.prev_offset)
            meta[key] = dict(ih=ih, iw=iw, H=H, W=W, fg=fg, mg=mg, bg=bg, kw=kw, frame_idx=3)
    # tracker persistence: three consecutive calls WITHOUT reset (module singleton semantics)
    rl.reset_state()
    offs = []
    for idx in range(3):
        bgr, d = synth.synth_frame(idx, 96, 160)
        L, R = r.pixel_shift_cuda(r.frame_to_tensor(bgr), torch.from_numpy(d)[None], 160, 96, 10.0, -2.5, -5.0,
                                  return_shift_map=False)
        offs.append(r.floating_window_tracker.prev_offset)
        out[f"seq{idx}__L"], out[f"seq{idx}__R"] = L, R
    out["seq_prev_offsets"] = np.array(offs, np.float64)
    out["meta_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    save("pixel_shift_cases.npz", **out)


# ------------------------------------------------------------------------------------------
# 3. helper functions / trackers
# ------------------------------------------------------------------------------------------
def gen_helpers():
    out = {}
    rng = np.random.default_rng(20250905)
    # estimate_subject_depth: normal, <20 valid, value==1.0, tie
    planes = {
        "ramp": synth.synth_frame(0, 60, 100)[1],
        "few_valid": np.full((60, 100), 0.99, np.float32),
        "ones_edge": np.clip(rng.random((60, 100)).astype(np.float32) * 1.2, 0, 1),
        "two_peaks": np.where(np.mgrid[0:60, 0:100][1] % 2 == 0, np.float32(0.30), np.float32(0.70)).astype(np.float32),
        "const_mid": np.full((60, 100), 0.5, np.float32),
    }
    for k, p in planes.items():
        out[f"subj_in__{k}"] = p
        out[f"subj_out__{k}"] = np.float32(r.estimate_subject_depth(torch.from_numpy(p)[None]).item())
    # quantiles incl. the float32-rank case
    for n in (1000, 36864, 518400):
        v = rng.random(n).astype(np.float32)
        out[f"quant_in__{n}"] = v if n <= 36864 else np.zeros(0, np.float32)
        out[f"quant_seed__{n}"] = np.int64(n)
        if n > 36864:
            v = np.random.default_rng(n).random(n).astype(np.float32)
        out[f"quant_out__{n}"] = np.array([torch.quantile(torch.from_numpy(v), q).item() for q in (0.02, 0.05, 0.5, 0.95, 0.98)],
                                          np.float32)
    # DepthPercentileEMA over a sequence incl. a collapsed (constant) frame; TemporalDepthFilter before it
    rl.reset_state()
    tdf = r.TemporalDepthFilter(alpha=0.5)
    seq = [synth.synth_frame(i, 54, 96)[1] for i in range(4)]
    seq.insert(2, np.full((54, 96), 0.4, np.float32))
    ema_state = []
    for i, d in enumerate(seq):
        dt = torch.from_numpy(d.copy())[None]
        f = tdf.smooth(dt)
        n = r.depth_ema_norm.normalize(f)
        out[f"ema_seq_filtered__{i}"] = f.numpy().copy()
        out[f"ema_seq_norm__{i}"] = n.numpy().copy()
        ema_state.append([float(r.depth_ema_norm._lo), float(r.depth_ema_norm._hi)])
        out[f"ema_seq_dyn__{i}"] = np.float64(r.compute_dynamic_parallax_scale(n, 0.90, 1.15))
    out["ema_seq_state"] = np.array(ema_state, np.float64)
    # constant frame FIRST (collapse with no state), then a normal one
    rl.reset_state()
    n0 = r.depth_ema_norm.normalize(torch.full((1, 54, 96), 0.4))
    out["ema_collapse_first"] = n0.numpy()
    out["ema_collapse_first_state_none"] = np.int64(r.depth_ema_norm._lo is None)
    # trackers
    fw = r.FloatingWindowTracker(alpha=0.97)
    xs = [0.01, 0.0105, 0.02, -0.01] + list(np.linspace(-0.3, 0.3, 120))
    out["fw_in"] = np.array(xs, np.float64)
    out["fw_out"] = np.array([fw.smooth_offset(float(v), threshold=0.0015) for v in xs], np.float64)
    ft_ = r.FocalDepthTracker(alpha=0.15, deadband=0.03, max_step=0.02)
    cands = [0.5, 0.52, 0.7, 0.7, 0.2, 0.21, 0.9, 0.05, 0.5, 0.5]
    mots = [0.0, 0.1, 0.5, 1.2, -0.3, 0.2, 0.7, 0.0, 0.3, 0.9]
    fo = []
    for c, m in zip(cands, mots):
        ft_.set_scene_motion(m)
        fo.append(ft_.update(c))
    out["focal_cand"], out["focal_motion"], out["focal_out"] = np.array(cands), np.array(mots), np.array(fo, np.float64)
    ce = r.ConvergenceEMA(alpha=0.97)
    cx = list(np.linspace(-0.01, 0.02, 12))
    out["conv_in"] = np.array(cx, np.float64)
    out["conv_out"] = np.array([ce.update(float(v)) for v in cx], np.float64)
    be = r.FloatingBarEaser(alpha=0.85)
    bx = [40, 40, 40, 0, 120, 7, 7, 7]
    out["bar_in"] = np.array(bx, np.int64)
    out["bar_out"] = np.array([be.ease(v) for v in bx], np.int64)
    ss = r.ShiftSmoother(0.15)
    so = [ss.smooth(4.5, -1.5, -6.0) for _ in range(4)]
    out["smoother_out"] = np.array(so, np.float64)
    # motion metric
    a, b = seq[0], seq[1]
    out["motion_out"] = np.float64(r.compute_motion_metric(torch.from_numpy(a)[None], torch.from_numpy(b)[None]))
    # color grade / dof on a small eye
    bgr, d = synth.synth_frame(2, 72, 128)
    t = r.frame_to_tensor(bgr)
    out["grade_identity"] = r.tensor_to_frame(r.apply_color_grade(t, 1.0, 1.0, 0.0))
    out["grade_strong"] = r.tensor_to_frame(r.apply_color_grade(t, 1.35, 1.10, 0.04))
    for ms in (2.0, 1.0, 3.3):
        o = r.apply_dof_cuda(t, torch.from_numpy(d)[None], 0.37, max_sigma=ms, focus_width=0.35)
        out[f"dof_ms{ms}"] = o.numpy()
    for k, sg in ((3, 0.5), (5, 1.0), (7, 1.5), (9, 2.0)):
        out[f"gauss_k{k}"] = ref_stubs._gaussian_kernel1d(k, sg).numpy()
    # sharpen / INTER_AREA / formats (OpenCV semantics restated in ref_stubs: unpinned)
    out["sharp_0.15"] = r.apply_sharpening(bgr, 0.15)
    out["sharp_0.2"] = r.apply_sharpening(bgr, 0.2)
    import cv2
    out["area_half_w"] = cv2.resize(bgr, (64, 72), interpolation=cv2.INTER_AREA)
    out["area_2x2"] = cv2.resize(bgr, (64, 36), interpolation=cv2.INTER_AREA)
    bgr2 = synth.synth_frame(5, 72, 128)[0]
    for fmt in ("Half-SBS", "Full-SBS", "Red-Cyan Anaglyph", "Passive Interlaced"):
        out[f"fmt__{fmt}"] = r.format_3d_output(bgr, bgr2, fmt)
    out["side_left_7"] = r.apply_side_mask(bgr, side="left", width=7)
    out["side_right_0"] = r.apply_side_mask(bgr, side="right", width=0)
    out["side_right_9"] = r.apply_side_mask(bgr, side="right", width=9)
    out["pad_wide"] = r.pad_to_aspect_ratio(synth.synth_frame(1, 54, 128)[0], 128, 72)
    save("helpers.npz", **out)


# ------------------------------------------------------------------------------------------
# 4. the real render_sbs_3d loop driven through a fake VideoCapture / VideoWriter
# ------------------------------------------------------------------------------------------
class _Aspect:
    def __init__(self, label):
        self.label = label

    def get(self):
        return self.label


LOOP_CASES = {
    # name: (src_h, src_w, n_frames, kwargs for render_sbs_3d)
    "half_sbs_cli": (108, 192, 6, dict(output_format="Half-SBS", output_height=108, fg_shift=10.0, mg_shift=-2.5,
                                       bg_shift=-5.0, sharpness_factor=0.15, dof_strength=2.0, feather_strength=10.0,
                                       blur_ksize=9, use_subject_tracking=True, use_floating_window=True)),
    "half_sbs_gui_nodof": (108, 192, 5, dict(output_format="Half-SBS", output_height=108, fg_shift=4.5, mg_shift=-1.5,
                                            bg_shift=-6.0, sharpness_factor=0.2, dof_strength=0.0, feather_strength=0.0,
                                            blur_ksize=1, use_subject_tracking=True, use_floating_window=True,
                                            zero_parallax_strength=0.01, color_saturation=1.2, color_contrast=1.05,
                                            color_brightness=0.02, ipd_factor=1.0)),
    "full_sbs_preserve": (90, 160, 4, dict(output_format="Full-SBS", output_height=90, fg_shift=10.0, mg_shift=-2.5,
                                           bg_shift=-5.0, sharpness_factor=0.15, dof_strength=1.5, feather_strength=10.0,
                                           blur_ksize=9, use_subject_tracking=True, use_floating_window=False,
                                           preserve_original_aspect=True, original_video_width=160,
                                           original_video_height=90, ipd_factor=0.0)),
    "interlaced": (90, 160, 3, dict(output_format="Passive Interlaced", output_height=90, fg_shift=10.0, mg_shift=-2.5,
                                    bg_shift=-5.0, sharpness_factor=0.15, dof_strength=2.0, feather_strength=10.0,
                                    blur_ksize=9, use_subject_tracking=False, use_floating_window=False)),
    "anaglyph_43crop": (120, 160, 3, dict(output_format="Red-Cyan Anaglyph", output_height=90, fg_shift=10.0,
                                          mg_shift=-2.5, bg_shift=-5.0, sharpness_factor=0.15, dof_strength=2.0,
                                          feather_strength=10.0, blur_ksize=9, use_subject_tracking=True,
                                          use_floating_window=True, ipd_factor=1.1)),
}


# DOF strengths beyond the defaults (round 4): 0.7 / 2.1 / 4.2 are strengths where MKL's vsExp is NOT the rounded exponential on
# some (2.1, 4.2: on all four) blur levels; 3.0 and 5.0 (the GUI slider's maximum, 21-tap Gaussian) run Gaussians beyond the fused kernel's 9 taps
_DOF_COMMON = dict(output_format="Half-SBS", output_height=108, fg_shift=10.0, mg_shift=-2.5, bg_shift=-5.0, sharpness_factor=0.15,
                   feather_strength=10.0, blur_ksize=9, use_subject_tracking=True, use_floating_window=True)
LOOP_CASES_DOF = {f"half_sbs_dof{str(s).replace('.', 'p')}": (108, 192, 3, dict(_DOF_COMMON, dof_strength=s)) for s in (0.7, 2.1, 3.0, 4.2, 5.0)}
LOOP_CASES_DOF["anaglyph_dof3p0"] = (120, 160, 2, dict(_DOF_COMMON, output_format="Red-Cyan Anaglyph", output_height=90, dof_strength=3.0))
LOOP_CASES_DOF["full_sbs_dof2p6"] = (90, 160, 2, dict(_DOF_COMMON, output_format="Full-SBS", output_height=90, dof_strength=2.6,
                                                      preserve_original_aspect=True, original_video_width=160, original_video_height=90))


def run_loop(name, reset=True, cases=None):
    sh, sw, n, kw = (cases or LOOP_CASES)[name]
    frames, depths = synth.synth_clip(n, sh, sw)
    ref_stubs._Clip.clips["in.mp4"] = frames
    ref_stubs._Clip.clips["depth.mp4"] = [synth.depth_to_u8_bgr(d) for d in depths]
    if reset:
        rl.reset_state()
    kw = dict(kw)
    args = dict(input_path="in.mp4", depth_path="depth.mp4", output_path="out.avi", selected_codec="XVID", fps=24.0,
                output_width=sw, selected_aspect_ratio=_Aspect("Default (16:9)"), aspect_ratios=r.aspect_ratios,
                suspend_flag=threading.Event(), cancel_flag=threading.Event())
    args.update(kw)
    with contextlib.redirect_stdout(io.StringIO()) as so:
        r.render_sbs_3d(**args)
    if "crashed" in so.getvalue():
        raise RuntimeError(so.getvalue())
    return ref_stubs._Clip.written["out.avi"]


def gen_loops():
    out = {}
    for name in LOOP_CASES:
        written = run_loop(name)
        out[f"{name}__frames"] = np.stack(written)
        print(f"  loop {name}: {len(written)} frames of {written[0].shape}")
    # singleton leak: a second render in the same process WITHOUT resetting the module singletons
    run_loop("half_sbs_cli", reset=True)
    w2 = run_loop("half_sbs_cli", reset=False)
    out["half_sbs_cli__second_render_frames"] = np.stack(w2)
    out["cases_json"] = np.frombuffer(json.dumps(LOOP_CASES).encode(), dtype=np.uint8)
    save("render_loop.npz", **out)


def gen_dof():
    out = {}
    for name in LOOP_CASES_DOF:
        written = run_loop(name, cases=LOOP_CASES_DOF)
        out[f"{name}__frames"] = np.stack(written)
        print(f"  dof loop {name}: {len(written)} frames of {written[0].shape}")
    out["cases_json"] = np.frombuffer(json.dumps(LOOP_CASES_DOF).encode(), dtype=np.uint8)
    save("dof_levels.npz", **out)


# ------------------------------------------------------------------------------------------
# 5. round-1 widening: fractional INTER_AREA (VR / non-preserve Full-SBS fit), black-bar auto crop
# ------------------------------------------------------------------------------------------
LOOP_CASES2 = {
    # VR: eye 1440x810 -> warp 1920x1080 -> pad_to_aspect_ratio(1440,1600) = INTER_AREA 4/3 + letterbox; frames are
    # 2880x1600, only bands are stored (rows 393..441 = top edge of the picture, rows 1200..1210 = bottom edge)
    "vr_1080": (405, 720, 3, dict(output_format="VR", output_height=1080, fg_shift=10.0, mg_shift=-2.5, bg_shift=-5.0,
                                sharpness_factor=0.15, dof_strength=2.0, feather_strength=10.0, blur_ksize=9,
                                use_subject_tracking=True, use_floating_window=True), [(393, 441), (1200, 1210)]),
    # letterboxed source (bars: 10 rows top, 14 rows bottom, value <= 9), auto crop + 2.0 -> 16:9 column crop
    "autocrop_letterbox": (120, 192, 5, dict(output_format="Half-SBS", output_height=108, fg_shift=10.0, mg_shift=-2.5,
                                             bg_shift=-5.0, sharpness_factor=0.15, dof_strength=2.0, feather_strength=10.0,
                                             blur_ksize=9, use_subject_tracking=True, use_floating_window=True,
                                             auto_crop_black_bars=True), None),
}


def run_loop2(name):
    sh, sw, n, kw, _ = LOOP_CASES2[name]
    if name == "autocrop_letterbox":
        frames, depth_bgr = synth.letterbox_clip(n, sh, sw, 10, 14)
    else:
        frames, depths = synth.synth_clip(n, sh, sw)
        depth_bgr = [synth.depth_to_u8_bgr(d) for d in depths]
    ref_stubs._Clip.clips["in.mp4"] = frames
    ref_stubs._Clip.clips["depth.mp4"] = depth_bgr
    rl.reset_state()
    args = dict(input_path="in.mp4", depth_path="depth.mp4", output_path="out.avi", selected_codec="XVID", fps=24.0,
                output_width=sw, selected_aspect_ratio=_Aspect("Default (16:9)"), aspect_ratios=r.aspect_ratios,
                suspend_flag=threading.Event(), cancel_flag=threading.Event())
    args.update(kw)
    with contextlib.redirect_stdout(io.StringIO()) as so:
        r.render_sbs_3d(**args)
    if "crashed" in so.getvalue():
        raise RuntimeError(so.getvalue())
    return ref_stubs._Clip.written["out.avi"], so.getvalue()


def gen_widen():
    import cv2
    out = {}
    # helper level: fractional / mixed INTER_AREA and pad_to_aspect_ratio at small sizes
    bgr = synth.synth_frame(3, 72, 128)[0]
    for nm, (dw, dh) in {"area_4_3": (96, 54), "area_3_2": (85, 48), "area_2p5": (51, 28), "area_mixed": (64, 54),
                         "area_1p07": (120, 67)}.items():
        out[nm] = cv2.resize(bgr, (dw, dh), interpolation=cv2.INTER_AREA)
    for nm, (dw, dh) in {"area_up_1p5": (192, 108), "area_up_quirk": (191, 108), "area_up_mixed": (256, 60), "area_up_y": (100, 90)}.items():
        out[nm] = cv2.resize(bgr, (dw, dh), interpolation=cv2.INTER_AREA)     # OpenCV: linear machinery, area-mode coefficients
    out["pad_up_720_like"] = r.pad_to_aspect_ratio(bgr, 192, 108)   # the 1280x720 -> 1920x1080 Full-SBS case in miniature
    out["pad_frac_wide"] = r.pad_to_aspect_ratio(bgr, 96, 60)     # 4/3 down + vertical bars
    out["pad_frac_tall"] = r.pad_to_aspect_ratio(bgr, 100, 48)    # 3/2 down + horizontal bars
    out["fmt__VR_identity"] = r.format_3d_output(np.zeros((1600, 1440, 3), np.uint8) + 7, np.zeros((1600, 1440, 3), np.uint8) + 9, "VR")[::200, ::240]
    # detect_black_bars on the letterboxed frames
    lf, _ = synth.letterbox_clip(4, 120, 192, 10, 14)
    out["bars"] = np.array([r.detect_black_bars(r.frame_to_tensor(f)) for f in lf], dtype=np.int32)
    out["bars_none"] = np.array(r.detect_black_bars(r.frame_to_tensor(np.zeros((40, 64, 3), np.uint8) + 200)), dtype=np.int32)
    out["bars_all_black"] = np.array(r.detect_black_bars(r.frame_to_tensor(np.zeros((40, 64, 3), np.uint8) + 4)), dtype=np.int32)
    for name, (sh, sw, n, kw, bands) in LOOP_CASES2.items():
        written, log = run_loop2(name)
        fr = np.stack(written)
        print(f"  loop {name}: {len(written)} frames of {written[0].shape}")
        out[f"{name}__shape"] = np.array(fr.shape, dtype=np.int64)
        if bands is None:
            out[f"{name}__frames"] = fr
        else:
            for (a, b) in bands:
                out[f"{name}__rows_{a}_{b}"] = fr[:, a:b]
            out[f"{name}__colsum"] = fr.astype(np.int64).sum(axis=(1, 3))   # per-frame per-column sums (coarse full-frame check)
        if name == "autocrop_letterbox":
            out[f"{name}__log"] = np.frombuffer(log.encode(), dtype=np.uint8)
    out["cases_json"] = np.frombuffer(json.dumps({k: list(v) for k, v in LOOP_CASES2.items()}).encode(), dtype=np.uint8)
    save("widen.npz", **out)


# ------------------------------------------------------------------------------------------
# 6. preview visualisers: core/preview_utils.py:generate_preview_image, the exactly defined types
# ------------------------------------------------------------------------------------------
PREVIEW_TYPES = ["Passive Interlaced", "HSBS", "Left-Right Diff", "Feather Blend", "Red-Blue Anaglyph"]


def gen_previews():
    pu = rl.load_preview_utils()
    out = {}
    for tag, (h, w) in {"even": (54, 96), "odd": (37, 75)}.items():
        left = synth.synth_frame(1, h, w)[0]
        right = synth.synth_frame(2, h, w)[0]
        shift = torch.zeros(1, h, w)
        for pt in PREVIEW_TYPES:
            out[f"{tag}__{pt}"] = pu.generate_preview_image(pt, left, right, shift, w, h)
    save("previews.npz", **out)
    # the colour-mapped types whose index arithmetic is plain numpy (:52-66); cv2.normalize (the two min-max heat-maps) is not stubbed.
    # applyColorMap is ref_stubs' table look-up through a synthetic table, stored next to the outputs.
    out = {"lut_JET": ref_stubs.test_colormap(ref_stubs.COLORMAP_JET), "lut_BONE": ref_stubs.test_colormap(ref_stubs.COLORMAP_BONE)}
    rng = np.random.default_rng(77)
    for tag, (h, w) in {"even": (54, 96), "odd": (37, 75)}.items():
        shift = (rng.standard_normal((1, h, w)) * 3.5).astype(np.float32)
        shift[0, 0, :8] = [-5.0, 5.0, 0.0, -0.0, 4.9999995, -7.5, 5.1, 0.02]
        out[f"{tag}__shift"] = shift
        left = synth.synth_frame(1, h, w)[0]
        for pt in ("Shift Heatmap (Clipped \u00b15px)", "Feather Mask"):
            out[f"{tag}__{pt}"] = pu.generate_preview_image(pt, left, left, torch.from_numpy(shift), w, h)
    save("previews_heat.npz", **out)


# ------------------------------------------------------------------------------------------
# 7. skip_blank_frames: the real loop with the blackdetect side channel replaced by a fixed list (ffmpeg is absent here;
#    the list is what detect_black_white_frames would return -- absolute frame indices, compared against loop index + start)
# ------------------------------------------------------------------------------------------
_CLI = dict(fg_shift=10.0, mg_shift=-2.5, bg_shift=-5.0, sharpness_factor=0.15, dof_strength=2.0, feather_strength=10.0,
            blur_ksize=9, use_subject_tracking=True, use_floating_window=True, skip_blank_frames=True)
BLANK_CASES = {
    # name: (src_h, src_w, n_frames, blank list, kwargs)
    "blank_half_sbs": (108, 192, 8, [1, 3, 4], dict(_CLI, output_format="Half-SBS", output_height=108, ipd_factor=1.1,
                                                    color_saturation=1.2, color_contrast=1.05, color_brightness=0.02)),
    # source 160x90 rendered at 192x108: the blank frame stays 160x90 and pad_to_aspect_ratio up-scales it by 1.2
    "blank_interlaced_up": (90, 160, 6, [0, 2], dict(_CLI, output_format="Passive Interlaced", output_height=108)),
    # 4:3 source: the blank frame is the UNCROPPED 160x120 frame -> INTER_AREA 4/3 down + pillar bars
    "blank_anaglyph_43": (120, 160, 6, [2, 3, 5], dict(_CLI, output_format="Red-Cyan Anaglyph", output_height=90, ipd_factor=0.9)),
}


def run_blank_loop(name):
    sh, sw, n, blank, kw = BLANK_CASES[name]
    frames, depths = synth.synth_clip(n, sh, sw)
    ref_stubs._Clip.clips["in.mp4"] = frames
    ref_stubs._Clip.clips["depth.mp4"] = [synth.depth_to_u8_bgr(d) for d in depths]
    rl.reset_state()
    args = dict(input_path="in.mp4", depth_path="depth.mp4", output_path="out.avi", selected_codec="XVID", fps=24.0,
                output_width=sw, selected_aspect_ratio=_Aspect("Default (16:9)"), aspect_ratios=r.aspect_ratios,
                suspend_flag=threading.Event(), cancel_flag=threading.Event())
    args.update(kw)
    saved = r.detect_black_white_frames
    r.detect_black_white_frames = lambda *a, **k: list(blank)
    try:
        with contextlib.redirect_stdout(io.StringIO()) as so:
            r.render_sbs_3d(**args)
    finally:
        r.detect_black_white_frames = saved
    if "crashed" in so.getvalue():
        raise RuntimeError(so.getvalue())
    assert so.getvalue().count("Skipping blank frame") == len([b for b in blank if b < n - 1]), so.getvalue()
    return ref_stubs._Clip.written["out.avi"]


def gen_blank():
    out = {}
    for name in BLANK_CASES:
        written = run_blank_loop(name)
        out[f"{name}__frames"] = np.stack(written)
        print(f"  blank loop {name}: {len(written)} frames of {written[0].shape}")
    out["cases_json"] = np.frombuffer(json.dumps({k: list(v) for k, v in BLANK_CASES.items()}).encode(), dtype=np.uint8)
    save("blank.npz", **out)


# ------------------------------------------------------------------------------------------
# 9. B2 attribution: the reference's OWN stage outputs inside the loop (eyes out of pixel_shift_cuda, the normalised depth,
#    the focal depth, the side-mask decision) next to its final muxed frame, so that the finishing stage (DOF + grade + bars +
#    sharpen + fit + mux) can be checked in isolation: feed the reference's eyes, compare with the reference's frame.
# ------------------------------------------------------------------------------------------
ATTRIB_CASES = {
    "half_sbs_cli": LOOP_CASES["half_sbs_cli"],
    "half_sbs_graded": (108, 192, 4, dict(output_format="Half-SBS", output_height=108, fg_shift=10.0, mg_shift=-2.5, bg_shift=-5.0,
                                         sharpness_factor=0.15, dof_strength=2.0, feather_strength=10.0, blur_ksize=9,
                                         use_subject_tracking=True, use_floating_window=True, color_saturation=1.2,
                                         color_contrast=1.05, color_brightness=0.02)),
    "full_sbs_preserve": LOOP_CASES["full_sbs_preserve"],
    "interlaced": LOOP_CASES["interlaced"],
    "anaglyph_43crop": LOOP_CASES["anaglyph_43crop"],
}


def run_loop_capturing(sh, sw, n, kw, depth_as_u8=True):
    """render_sbs_3d through the fake cv2 with taps on pixel_shift_cuda / FocalDepthTracker.update / compute_motion_metric /
    apply_side_mask.  Returns (written frames, per-frame captures)."""
    frames, depths = synth.synth_clip(n, sh, sw)
    ref_stubs._Clip.clips["in.mp4"] = frames
    ref_stubs._Clip.clips["depth.mp4"] = [synth.depth_to_u8_bgr(d) for d in depths]
    rl.reset_state()
    caps = []
    cur = {}
    orig_ps, orig_mm, orig_fu, orig_sm = r.pixel_shift_cuda, r.compute_motion_metric, r.FocalDepthTracker.update, r.apply_side_mask

    def ps(*a, **k):
        out = orig_ps(*a, **k)
        cur.clear()
        cur["L"], cur["R"] = np.array(out[0], copy=True), np.array(out[1], copy=True)
        cur["side"], cur["bar"] = 0, 0
        caps.append(cur.copy())
        return out

    def mm(prev, d):
        caps[-1]["dn"] = d.detach().cpu().numpy().copy()
        return orig_mm(prev, d)

    def fu(self_, cand):
        v = orig_fu(self_, cand)
        caps[-1]["focal"] = float(v)
        return v

    def sm(img, side="right", width=40):
        caps[-1]["side"], caps[-1]["bar"] = (1 if side == "right" else 2), int(width)
        return orig_sm(img, side=side, width=width)

    r.pixel_shift_cuda, r.compute_motion_metric, r.FocalDepthTracker.update, r.apply_side_mask = ps, mm, fu, sm
    try:
        args = dict(input_path="in.mp4", depth_path="depth.mp4", output_path="out.avi", selected_codec="XVID", fps=24.0,
                    output_width=sw, selected_aspect_ratio=_Aspect("Default (16:9)"), aspect_ratios=r.aspect_ratios,
                    suspend_flag=threading.Event(), cancel_flag=threading.Event())
        args.update(kw)
        with contextlib.redirect_stdout(io.StringIO()) as so:
            r.render_sbs_3d(**args)
        if "crashed" in so.getvalue():
            raise RuntimeError(so.getvalue())
    finally:
        r.pixel_shift_cuda, r.compute_motion_metric, r.FocalDepthTracker.update, r.apply_side_mask = orig_ps, orig_mm, orig_fu, orig_sm
    return ref_stubs._Clip.written["out.avi"], caps


def gen_attrib():
    out = {"cases_json": np.frombuffer(json.dumps(ATTRIB_CASES).encode(), dtype=np.uint8)}
    for name, (sh, sw, n, kw) in ATTRIB_CASES.items():
        written, caps = run_loop_capturing(sh, sw, n, kw)
        assert len(written) == len(caps)
        out[f"{name}__final"] = np.stack(written)
        out[f"{name}__L"] = np.stack([c["L"] for c in caps])
        out[f"{name}__R"] = np.stack([c["R"] for c in caps])
        out[f"{name}__dn"] = np.stack([c["dn"] for c in caps])
        out[f"{name}__focal"] = np.array([c["focal"] for c in caps], np.float64)
        out[f"{name}__bar"] = np.array([[c["side"], c["bar"]] for c in caps], np.int32)
        print(f"  attrib {name}: {len(written)} frames, eyes {caps[0]['L'].shape}, dn {caps[0]['dn'].shape}")
    save("attrib.npz", **out)


# ------------------------------------------------------------------------------------------
# 10. BASELINE configs[0] at REAL size: one 1920x1080 clip (3 decoded frames -> 2 rendered), CLI defaults + DOF 2.0, Half-SBS.
#     The muxed frames are 6.2 MB each, so the fixture keeps row bands + per-row / per-column channel sums (like widen.npz).
# ------------------------------------------------------------------------------------------
REAL_KW = dict(output_format="Half-SBS", output_height=1080, fg_shift=10.0, mg_shift=-2.5, bg_shift=-5.0, sharpness_factor=0.15,
               dof_strength=2.0, feather_strength=10.0, blur_ksize=9, use_subject_tracking=True, use_floating_window=True)
REAL_BANDS = [(0, 8), (268, 276), (536, 544), (804, 812), (1072, 1080)]


def gen_real1080():
    sh, sw, n = 1080, 1920, 3
    written, _ = run_loop_capturing(sh, sw, n, REAL_KW)
    out = {"kw_json": np.frombuffer(json.dumps(REAL_KW).encode(), dtype=np.uint8),
           "bands_json": np.frombuffer(json.dumps(REAL_BANDS).encode(), dtype=np.uint8)}
    for i, fr in enumerate(written):
        assert fr.shape == (1080, 1920, 3)
        out[f"bands_{i}"] = np.concatenate([fr[a:b] for a, b in REAL_BANDS])
        out[f"rowsum_{i}"] = fr.astype(np.int64).sum(axis=1)      # [1080, 3]
        out[f"colsum_{i}"] = fr.astype(np.int64).sum(axis=0)      # [1920, 3]
        out[f"sha_{i}"] = np.frombuffer(sha(fr).encode(), dtype=np.uint8)
        # a 1-in-8 decimated copy: every 8th row and column, all channels (16 KB) -- localises a difference the sums only detect
        out[f"dec8_{i}"] = fr[::8, ::8].copy()
    print(f"  real1080: {len(written)} frames")
    save("real1080.npz", **out)


# ------------------------------------------------------------------------------------------
# 11. the other output formats at REAL size: Full-SBS with preserve_original_aspect (eye = the 1920x1080 frame
#     itself, identity resize: the most noise-sensitive case), Passive Interlaced, Red-Cyan Anaglyph.  Same storage as real1080.npz.
# ------------------------------------------------------------------------------------------
# 10b. BASELINE configs[2] at REAL size (round 4): one 3840x2160 clip (3 decoded frames -> 2 rendered), CLI defaults + DOF 2.0, Half-SBS at the
#      source size.  24.9 MB per muxed frame: row bands, an 8x decimated copy, per-row / per-column sums and the SHA-256 of the whole frame.
#      (~50 s of reference time on 8 threads.)
# ------------------------------------------------------------------------------------------
REAL4K_KW = dict(REAL_KW, output_height=2160)
REAL4K_BANDS = [(0, 8), (536, 544), (1076, 1084), (1616, 1624), (2152, 2160)]


def gen_real4k():
    sh, sw, n = 2160, 3840, 3
    written, _ = run_loop_capturing(sh, sw, n, REAL4K_KW)
    out = {"kw_json": np.frombuffer(json.dumps(REAL4K_KW).encode(), dtype=np.uint8),
           "bands_json": np.frombuffer(json.dumps(REAL4K_BANDS).encode(), dtype=np.uint8)}
    for i, fr in enumerate(written):
        assert fr.shape == (2160, 3840, 3)
        out[f"bands_{i}"] = np.concatenate([fr[a:b] for a, b in REAL4K_BANDS])
        out[f"rowsum_{i}"] = fr.astype(np.int64).sum(axis=1)
        out[f"colsum_{i}"] = fr.astype(np.int64).sum(axis=0)
        out[f"sha_{i}"] = np.frombuffer(sha(fr).encode(), dtype=np.uint8)
        out[f"dec8_{i}"] = fr[::8, ::8].copy()
    print(f"  real4k: {len(written)} frames")
    save("real4k.npz", **out)


# ------------------------------------------------------------------------------------------
_REAL_COMMON = dict(output_height=1080, fg_shift=10.0, mg_shift=-2.5, bg_shift=-5.0, sharpness_factor=0.15, dof_strength=2.0,
                    feather_strength=10.0, blur_ksize=9, use_subject_tracking=True, use_floating_window=True)
REAL_FORMAT_CASES = {
    "full_sbs_preserve": dict(_REAL_COMMON, output_format="Full-SBS", preserve_original_aspect=True, original_video_width=1920,
                              original_video_height=1080),
    "interlaced": dict(_REAL_COMMON, output_format="Passive Interlaced"),
    "anaglyph": dict(_REAL_COMMON, output_format="Red-Cyan Anaglyph"),
}


def gen_real1080_formats():
    sh, sw, n = 1080, 1920, 3
    out = {"cases_json": np.frombuffer(json.dumps(REAL_FORMAT_CASES).encode(), dtype=np.uint8),
           "bands_json": np.frombuffer(json.dumps(REAL_BANDS).encode(), dtype=np.uint8)}
    for name, kw in REAL_FORMAT_CASES.items():
        written, _ = run_loop_capturing(sh, sw, n, kw)
        out[f"{name}__shape"] = np.array(written[0].shape, dtype=np.int64)
        for i, fr in enumerate(written):
            out[f"{name}__bands_{i}"] = np.concatenate([fr[a:b] for a, b in REAL_BANDS])
            out[f"{name}__rowsum_{i}"] = fr.astype(np.int64).sum(axis=1)
            out[f"{name}__colsum_{i}"] = fr.astype(np.int64).sum(axis=0)
            out[f"{name}__dec8_{i}"] = fr[::8, ::8].copy()
            out[f"{name}__sha_{i}"] = np.frombuffer(sha(fr).encode(), dtype=np.uint8)
        print(f"  real1080 {name}: {len(written)} frames of {written[0].shape}")
    save("real1080_formats.npz", **out)


def gen_heal():
    """a23 heal_missing_pixels (core/render_3d.py:431-459), pure torch: pinned directly."""
    out = {"cases": np.frombuffer(json.dumps(HEAL_CASES).encode(), dtype=np.uint8)}
    for i, case in enumerate(HEAL_CASES):
        warped, orig, edge, hs = heal_inputs(case)
        res = r.heal_missing_pixels(torch.from_numpy(warped), None, torch.from_numpy(orig),
                                    None if edge is None else torch.from_numpy(edge), hs).numpy()
        out[f"healed_{i}"] = res
        out[f"sha_{i}"] = np.frombuffer(sha(res).encode(), dtype=np.uint8)
    save("heal.npz", **out)


# ------------------------------------------------------------------------------------------
# 11b. The other output formats at 3840x2160 (round 4): Full-SBS with preserve_original_aspect (eyes = the frame itself), Passive Interlaced,
#      Red-Cyan Anaglyph, VR (fractional 8/3 INTER_AREA into the 1440x1600 canvas).  Per frame: SHA-256 of the whole frame, row / column sums
#      (int32) and two 4-row bands -- 50 s of reference time per format.
# ------------------------------------------------------------------------------------------
_REAL4K_COMMON = dict(_REAL_COMMON, output_height=2160)
REAL4K_FORMAT_CASES = {
    "full_sbs_preserve": dict(_REAL4K_COMMON, output_format="Full-SBS", preserve_original_aspect=True, original_video_width=3840,
                              original_video_height=2160),
    "interlaced": dict(_REAL4K_COMMON, output_format="Passive Interlaced"),
    "anaglyph": dict(_REAL4K_COMMON, output_format="Red-Cyan Anaglyph"),
    "vr": dict(_REAL4K_COMMON, output_format="VR"),
}
REAL4K_FORMAT_BANDS = [(0, 4), (-4, None)]


# DOF strengths beyond the fused finishing kernel's 9 taps at 3840x2160: 3.0 (13 taps) Half-SBS and the slider's maximum 5.0 (21 taps) as anaglyph
REAL4K_DOF_CASES = {
    "half_sbs_dof3": dict(_REAL4K_COMMON, output_format="Half-SBS", dof_strength=3.0),
    "anaglyph_dof5": dict(_REAL4K_COMMON, output_format="Red-Cyan Anaglyph", dof_strength=5.0),
}


def gen_real4k_formats(cases=None, fname="real4k_formats.npz"):
    cases = REAL4K_FORMAT_CASES if cases is None else cases
    sh, sw, n = 2160, 3840, 3
    out = {"cases_json": np.frombuffer(json.dumps(cases).encode(), dtype=np.uint8)}
    for name, kw in cases.items():
        written, _ = run_loop_capturing(sh, sw, n, kw)
        out[f"{name}__shape"] = np.array(written[0].shape, dtype=np.int64)
        for i, fr in enumerate(written):
            out[f"{name}__bands_{i}"] = np.concatenate([fr[a:b] for a, b in REAL4K_FORMAT_BANDS])
            out[f"{name}__rowsum_{i}"] = fr.astype(np.int64).sum(axis=1).astype(np.int32)
            out[f"{name}__colsum_{i}"] = fr.astype(np.int64).sum(axis=0).astype(np.int32)
            out[f"{name}__sha_{i}"] = np.frombuffer(sha(fr).encode(), dtype=np.uint8)
        print(f"  real4k {name}: {len(written)} frames of {written[0].shape}")
    save(fname, **out)


# ------------------------------------------------------------------------------------------
# 12. Random render_sbs_3d configurations at 1920x1080 (round 4): every control the loop forwards drawn at random (the generator of
#     tests/test_oracle_vs_live_reference.py::test_render_loop_every_control_exact_on_untailed_planes), 12 configurations, two rendered frames
#     each.  Stored: the keyword set, SHA-256 and row sums of every frame (a few KB) -- the frames themselves are reproduced, not kept.
# ------------------------------------------------------------------------------------------
def random_controls(seed, sh, sw):
    rng = np.random.default_rng(9500 + seed)
    fmt = ["Half-SBS", "Full-SBS", "Passive Interlaced", "Red-Cyan Anaglyph", "VR"][int(rng.integers(0, 5))]
    rng.integers(0, 2)   # (the small-size sweep draws its frame size here)
    kw = dict(output_format=fmt, output_height=sh, fg_shift=float(rng.uniform(0, 30)), mg_shift=float(rng.uniform(-10, 5)),
              bg_shift=float(rng.uniform(-25, 0)), sharpness_factor=float(rng.uniform(0.0, 0.6)),
              dof_strength=float([0.0, 1.0, 2.0, 2.0, 3.3][int(rng.integers(0, 5))]), feather_strength=float(rng.uniform(0, 20)),
              blur_ksize=int(rng.integers(0, 7)) * 2 + 1, use_subject_tracking=bool(rng.integers(0, 2)),
              use_floating_window=bool(rng.integers(0, 2)), max_pixel_shift_percent=float(rng.uniform(0.005, 0.06)),
              zero_parallax_strength=float(rng.uniform(0, 0.03)) if rng.integers(0, 2) else 0.0,
              enable_edge_masking=bool(rng.integers(0, 3) > 0), enable_feathering=bool(rng.integers(0, 3) > 0),
              convergence_strength=float([0.0, 3.0, -2.0][int(rng.integers(0, 3))]), enable_dynamic_convergence=bool(rng.integers(0, 2)),
              ipd_factor=float([1.0, 0.0, 1.2, 0.8][int(rng.integers(0, 4))]),
              color_saturation=float(rng.uniform(0.8, 1.4)), color_contrast=float(rng.uniform(0.9, 1.2)),
              color_brightness=float(rng.uniform(-0.05, 0.05)))
    if fmt == "Full-SBS" or rng.integers(0, 3) == 0:
        kw.update(preserve_original_aspect=True, original_video_width=sw, original_video_height=sh)
    return kw


def gen_real1080_random(seeds=range(12), sh=1080, sw=1920, fname="real1080_random.npz"):
    n = 3
    cases = {f"seed{sd}": random_controls(sd, sh, sw) for sd in seeds}
    out = {"cases_json": np.frombuffer(json.dumps(cases).encode(), dtype=np.uint8)}
    for name, kw in cases.items():
        LOOP_CASES[name] = (sh, sw, n, kw)
        try:
            written = run_loop(name)
        finally:
            del LOOP_CASES[name]
        out[f"{name}__shape"] = np.array(written[0].shape, dtype=np.int64)
        for i, fr in enumerate(written):
            out[f"{name}__rowsum_{i}"] = fr.astype(np.int64).sum(axis=1).astype(np.int32)
            out[f"{name}__sha_{i}"] = np.frombuffer(sha(fr).encode(), dtype=np.uint8)
        print(f"  real1080 random {name}: {kw['output_format']}, {len(written)} frames of {written[0].shape}")
    save(fname, **out)


# ------------------------------------------------------------------------------------------
# 13. Black-bar auto crop at 1920x1080 (round 4): letterboxed clips (synth.letterbox_clip: dark bars with a little noise, one frame with a dark
#     first content row) through render_sbs_3d with auto_crop_black_bars -- detection per frame, crop, re-fit.  SHA-256 + row sums per frame.
# ------------------------------------------------------------------------------------------
_LB_BASE = dict(output_height=1080, fg_shift=10.0, mg_shift=-2.5, bg_shift=-5.0, sharpness_factor=0.15, dof_strength=2.0, feather_strength=10.0,
                blur_ksize=9, use_subject_tracking=True, use_floating_window=True, auto_crop_black_bars=True)
LETTERBOX_CASES = {
    "half_sbs_140_140": (140, 140, dict(_LB_BASE, output_format="Half-SBS")),
    "half_sbs_131_137": (131, 137, dict(_LB_BASE, output_format="Half-SBS")),
    "full_sbs_138_138": (138, 138, dict(_LB_BASE, output_format="Full-SBS", preserve_original_aspect=True, original_video_width=1920,
                                        original_video_height=1080)),
    "anaglyph_100_60": (100, 60, dict(_LB_BASE, output_format="Red-Cyan Anaglyph")),
}


def gen_real1080_letterbox():
    sh, sw, n = 1080, 1920, 3
    out = {"cases_json": np.frombuffer(json.dumps(LETTERBOX_CASES).encode(), dtype=np.uint8)}
    for name, (top, bottom, kw) in LETTERBOX_CASES.items():
        frames, depth_bgr = synth.letterbox_clip(n, sh, sw, top, bottom)
        ref_stubs._Clip.clips["in.mp4"] = frames
        ref_stubs._Clip.clips["depth.mp4"] = depth_bgr
        rl.reset_state()
        args = dict(input_path="in.mp4", depth_path="depth.mp4", output_path="out.avi", selected_codec="XVID", fps=24.0,
                    output_width=sw, selected_aspect_ratio=_Aspect("Default (16:9)"), aspect_ratios=r.aspect_ratios,
                    suspend_flag=threading.Event(), cancel_flag=threading.Event())
        args.update(kw)
        with contextlib.redirect_stdout(io.StringIO()) as so:
            r.render_sbs_3d(**args)
        if "crashed" in so.getvalue():
            raise RuntimeError(so.getvalue())
        written = ref_stubs._Clip.written["out.avi"]
        out[f"{name}__shape"] = np.array(written[0].shape, dtype=np.int64)
        for i, fr in enumerate(written):
            out[f"{name}__rowsum_{i}"] = fr.astype(np.int64).sum(axis=1).astype(np.int32)
            out[f"{name}__sha_{i}"] = np.frombuffer(sha(fr).encode(), dtype=np.uint8)
        print(f"  letterbox {name}: {len(written)} frames of {written[0].shape}")
    save("real1080_letterbox.npz", **out)


# ------------------------------------------------------------------------------------------
# 14. The GUI's OWN default configuration (round 5; VisionDepth3D.py:1405-1453: Full-SBS, fg 4.5 / mg -1.5 / bg -6, blur_ksize 1, feather_strength 0.0,
#     sharpness 0.2, zero_parallax 0.01, auto crop on) at real sizes: 1920x1080 (eyes at warp resolution), 3840x2160 (the 2 x 2 fit into 1920x1080 eyes) and
#     the Half-SBS variant.  feather_strength 0 is where the library skips the mask / window-sum / blend kernels (an exact no-op of feather_shift_edges),
#     so these frames pin that decision against the reference on the GPU box.  SHA-256 + row sums per frame; the fixture records torch's thread count
#     (torch.mean's summation order depends on it: vd3d_render_params::aten_sum_threads).
# ------------------------------------------------------------------------------------------
GUI_DEFAULTS = dict(output_format="Full-SBS", fg_shift=4.5, mg_shift=-1.5, bg_shift=-6.0, sharpness_factor=0.2, dof_strength=2.0, feather_strength=0.0,
                    blur_ksize=1, use_subject_tracking=True, use_floating_window=True, max_pixel_shift_percent=0.02, auto_crop_black_bars=True,
                    parallax_balance=0.8, zero_parallax_strength=0.01, enable_edge_masking=True, enable_feathering=True,
                    convergence_strength=0.0, enable_dynamic_convergence=True)
GUI_CASES = {
    "gui_1080_full": (1080, 1920, dict(GUI_DEFAULTS, output_height=1080)),
    "gui_1080_half": (1080, 1920, dict(GUI_DEFAULTS, output_height=1080, output_format="Half-SBS")),
    "gui_4k_full": (2160, 3840, dict(GUI_DEFAULTS, output_height=2160)),
    "gui_1080_blur9_feather0": (1080, 1920, dict(GUI_DEFAULTS, output_height=1080, blur_ksize=9, output_format="Half-SBS")),
}


def gen_gui_defaults():
    n = 4
    out = {"cases_json": np.frombuffer(json.dumps({k: [v[0], v[1], v[2]] for k, v in GUI_CASES.items()}).encode(), dtype=np.uint8),
           "torch_threads": np.array([torch.get_num_threads()], dtype=np.int64)}
    for name, (sh, sw, kw) in GUI_CASES.items():
        LOOP_CASES[name] = (sh, sw, n, kw)
        try:
            written = run_loop(name)
        finally:
            del LOOP_CASES[name]
        out[f"{name}__shape"] = np.array(written[0].shape, dtype=np.int64)
        for i, fr in enumerate(written):
            out[f"{name}__rowsum_{i}"] = fr.astype(np.int64).sum(axis=1).astype(np.int32)
            out[f"{name}__sha_{i}"] = np.frombuffer(sha(fr).encode(), dtype=np.uint8)
        print(f"  gui defaults {name}: {kw['output_format']}, {len(written)} frames of {written[0].shape}")
    save("gui_defaults.npz", **out)


def shift_sweep_case(seed):
    """One random `pixel_shift_cuda` configuration at an odd, small size (the body of tests/test_oracle_vs_live_reference.py::test_pixel_shift_random_parameters and of
    tools/sweep_live_any_size.py shift)."""
    rng = np.random.default_rng(5000 + seed)
    ih, iw = int(rng.integers(24, 80)), int(rng.integers(32, 130))
    H, W = (ih, iw) if rng.integers(0, 3) == 0 else (int(rng.integers(24, 120)), int(rng.integers(32, 200)))
    kw = dict(blur_ksize=int(rng.integers(0, 6)) * 2 + 1, feather_strength=float(rng.uniform(0, 20)),
              use_subject_tracking=bool(rng.integers(0, 2)), enable_floating_window=bool(rng.integers(0, 2)),
              max_pixel_shift_percent=float(rng.uniform(0.005, 0.06)), zero_parallax_strength=float(rng.uniform(0, 0.03)),
              enable_edge_masking=bool(rng.integers(0, 3) > 0), enable_feathering=bool(rng.integers(0, 3) > 0),
              convergence_strength=float([0.0, 3.0, -2.0][int(rng.integers(0, 3))]), enable_dynamic_convergence=bool(rng.integers(0, 2)),
              depth_pop_gamma=float(rng.uniform(0.6, 1.3)), depth_pop_mid=float(rng.uniform(0.35, 0.65)),
              parallax_balance=float(rng.uniform(0.5, 1.0)))
    fg, mg_, bg = float(rng.uniform(0, 30)), float(rng.uniform(-10, 5)), float(rng.uniform(-25, 0))
    return ih, iw, H, W, fg, mg_, bg, kw


# round 5: `pixel_shift_cuda` on SMALL, odd planes with a fixed torch thread count -- where ATen's scalar tails, its premultiplied-weight bilinear kernel and the fused
# `lerp` of torch.quantile all show (seed 970 is the 39 x 27 plane on which a 600-configuration sweep found the last of them): float32 shift map, both eyes, tracker
SMALL_PLANE_SEEDS = {970: 2, 3: 4, 7: 1, 12: 8, 21: 3, 101: 4, 159: 1, 594: 8}   # seed -> torch threads


def gen_pixel_shift_small():
    out, meta = {}, {}
    prev = torch.get_num_threads()
    try:
        for seed, threads in SMALL_PLANE_SEEDS.items():
            ih, iw, H, W, fg, mg_, bg, kw = shift_sweep_case(seed)
            torch.set_num_threads(threads)
            assert torch.get_num_threads() == threads
            bgr, d = synth.synth_frame(seed, ih, iw)
            rl.reset_state()
            with torch.no_grad():
                L, R, S = r.pixel_shift_cuda(r.frame_to_tensor(bgr), torch.from_numpy(d)[None], W, H, fg, mg_, bg, **kw)
            key = f"small{seed}_t{threads}"
            out[key + "__L"], out[key + "__R"] = np.asarray(L), np.asarray(R)
            out[key + "__S"] = S.numpy().astype(np.float32)
            out[key + "__prev_offset"] = np.float64(r.floating_window_tracker.prev_offset)
            meta[key] = dict(ih=ih, iw=iw, H=H, W=W, fg=fg, mg=mg_, bg=bg, kw=dict(kw, aten_threads=threads), frame_idx=seed)
            print(f"  small plane {key}: {iw}x{ih} -> {W}x{H}, {threads} torch threads")
    finally:
        torch.set_num_threads(prev)
    out["meta_json"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    save("pixel_shift_small_planes.npz", **out)


def any_size_case(seed):
    """One random configuration of the live any-size sweep (tests/test_oracle_vs_live_reference.py::test_render_loop_any_size_exact_in_aten_mode draws the same)."""
    rng = np.random.default_rng(9700 + seed)
    fmt = ["Half-SBS", "Full-SBS", "Passive Interlaced", "Red-Cyan Anaglyph", "VR"][int(rng.integers(0, 5))]
    sh, sw = (int(rng.integers(20, 76)) * 2, int(rng.integers(30, 131)) * 2) if seed % 2 == 0 else (int(rng.integers(40, 150)), int(rng.integers(60, 260)))
    kw = dict(output_format=fmt, output_height=sh, fg_shift=float(rng.uniform(0, 30)), mg_shift=float(rng.uniform(-10, 5)),
              bg_shift=float(rng.uniform(-25, 0)), sharpness_factor=float(rng.uniform(0.0, 0.6)),
              dof_strength=float([0.0, 1.0, 2.0, 2.0, 3.3][int(rng.integers(0, 5))]), feather_strength=float(rng.uniform(0, 20)),
              blur_ksize=int(rng.integers(0, 7)) * 2 + 1, use_subject_tracking=bool(rng.integers(0, 2)),
              use_floating_window=bool(rng.integers(0, 2)), max_pixel_shift_percent=float(rng.uniform(0.005, 0.06)),
              enable_edge_masking=bool(rng.integers(0, 3) > 0), enable_feathering=bool(rng.integers(0, 3) > 0),
              convergence_strength=float([0.0, 3.0, -2.0][int(rng.integers(0, 3))]), enable_dynamic_convergence=bool(rng.integers(0, 2)),
              ipd_factor=float([1.0, 0.0, 1.2, 0.8][int(rng.integers(0, 4))]),
              color_saturation=float(rng.uniform(0.8, 1.4)), color_contrast=float(rng.uniform(0.9, 1.2)),
              color_brightness=float(rng.uniform(-0.05, 0.05)))
    if rng.integers(0, 3) == 0:
        kw.update(preserve_original_aspect=True, original_video_width=sw, original_video_height=sh)
    return sh, sw, kw


# round 5: the reference's render loop at frame sizes NO size rule covers (odd widths / heights, sources the loop crops to 16:9, eyes of H + W <= 128), run with a
# FIXED number of torch threads per case -- the N-thread ATen mode (aten_sum_threads) must reproduce these frames on the oracle and through the C ABI
ANY_SIZE_SEEDS = {1: 4, 2: 1, 3: 8, 4: 3, 5: 1, 6: 4, 9: 2, 11: 8, 12: 4, 15: 1}   # seed -> torch threads


def gen_aten_any_size():
    n = 4
    prev = torch.get_num_threads()
    cases = {}
    out = {}
    try:
        for seed, threads in ANY_SIZE_SEEDS.items():
            sh, sw, kw = any_size_case(seed)
            name = f"any{seed}_t{threads}"
            torch.set_num_threads(threads)
            assert torch.get_num_threads() == threads
            LOOP_CASES[name] = (sh, sw, n, kw)
            try:
                written = run_loop(name)
            finally:
                del LOOP_CASES[name]
            cases[name] = [sh, sw, kw, threads]
            out[f"{name}__shape"] = np.array(written[0].shape, dtype=np.int64)
            for i, fr in enumerate(written):
                out[f"{name}__rowsum_{i}"] = fr.astype(np.int64).sum(axis=1).astype(np.int32)
                out[f"{name}__sha_{i}"] = np.frombuffer(sha(fr).encode(), dtype=np.uint8)
            print(f"  any size {name}: {sw}x{sh} {kw['output_format']}, {threads} torch threads, {len(written)} frames of {written[0].shape}")
    finally:
        torch.set_num_threads(prev)
    out["cases_json"] = np.frombuffer(json.dumps(cases).encode(), dtype=np.uint8)
    save("aten_any_size.npz", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["kat", "shift", "helpers", "loops", "widen", "previews", "blank", "heal", "attrib", "real1080", "real1080_formats", "real4k", "real4k_formats", "real4k_dof", "real1080_random", "real4k_random", "real1080_letterbox", "gui_defaults", "aten_any_size", "shift_small"]
    if "aten_any_size" in which:
        gen_aten_any_size()
    if "shift_small" in which:
        gen_pixel_shift_small()
    if "gui_defaults" in which:
        gen_gui_defaults()
    if "attrib" in which:
        gen_attrib()
    if "real1080" in which:
        gen_real1080()
    if "real4k" in which:
        gen_real4k()
    if "real4k_formats" in which:
        gen_real4k_formats()
    if "real4k_dof" in which:
        gen_real4k_formats(REAL4K_DOF_CASES, "real4k_dof.npz")
    if "real1080_random" in which:
        gen_real1080_random()
    if "real1080_letterbox" in which:
        gen_real1080_letterbox()
    if "real4k_random" in which:   # the same generator at 3840x2160: shifts up to 6 % of the width = 230 pixels, blur sizes up to 13
        gen_real1080_random(seeds=(0, 2, 3, 8, 13), sh=2160, sw=3840, fname="real4k_random.npz")
    if "real1080_formats" in which:
        gen_real1080_formats()
    if "heal" in which:
        gen_heal()
    if "blank" in which:
        gen_blank()
    if "previews" in which:
        gen_previews()
    if "widen" in which:
        gen_widen()
    if "kat" in which:
        gen_kat()
    if "shift" in which:
        gen_pixel_shift()
    if "helpers" in which:
        gen_helpers()
    if "loops" in which:
        gen_loops()
    if "dof" in which:
        gen_dof()
