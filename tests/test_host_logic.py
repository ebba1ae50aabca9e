import os
"""CPU-only tests of the host logic that mirrors the reference's Python: geometry, kwargs mapping, synth."""
import numpy as np
import pytest

from visiondepth3d_amd import synth
from visiondepth3d_amd._abi import FMT_HALF_SBS, ShiftParams
from visiondepth3d_amd.geometry import plan_geometry
from visiondepth3d_amd.params import render_kwargs_to_params, shift_params_from_kwargs


def test_half_sbs_geometry_1080p_and_4k():
    # 1080p -> eye 540x960 -> warp 1080x1920 -> fit 1080x960 -> hstack 1920x1080
    g = plan_geometry(1920, 1080, 1080, "Half-SBS")
    assert (g["eye_w"], g["eye_h"], g["warp_w"], g["warp_h"], g["fit_w"], g["fit_h"], g["out_w"], g["out_h"]) == \
        (960, 540, 1920, 1080, 960, 1080, 1920, 1080)
    g = plan_geometry(3840, 2160, 2160, "Half-SBS")
    assert (g["eye_w"], g["eye_h"], g["warp_w"], g["warp_h"], g["out_w"], g["out_h"]) == (1920, 1080, 3840, 2160, 3840, 2160)


def test_full_sbs_is_hardwired_1080p_unless_preserve():
    g = plan_geometry(3840, 2160, 2160, "Full-SBS")  # core/render_3d.py:1120-1123
    assert (g["fit_w"], g["fit_h"], g["out_w"], g["out_h"]) == (1920, 1080, 3840, 1080)
    g = plan_geometry(3840, 2160, 2160, "Full-SBS", preserve_original_aspect=True, original_video_width=3840, original_video_height=2160)
    assert (g["eye_w"], g["eye_h"], g["out_w"], g["out_h"]) == (3840, 2160, 7680, 2160)


def test_centre_crop_when_aspect_differs():
    g = plan_geometry(160, 120, 90, "Half-SBS")  # 4:3 source, 16:9 target
    assert (g["crop_x"], g["crop_y"], g["crop_w"], g["crop_h"]) == (0, 15, 160, 90)
    g = plan_geometry(400, 100, 90, "Half-SBS")  # wider than target
    assert g["crop_w"] == int(100 * 16 / 9) and g["crop_x"] == (400 - g["crop_w"]) // 2


def test_render_kwargs_mirror_reference_forwarding():
    p = render_kwargs_to_params(192, 108, output_height=108, fg_shift=10.0, mg_shift=-2.5, bg_shift=-5.0, sharpness_factor=0.15,
                                output_format="Half-SBS", dof_strength=2.0, depth_pop_gamma=0.5, parallax_balance=0.3,
                                use_subject_tracking=True, use_floating_window=True)
    # accepted but never forwarded by render_sbs_3d (:1284-1331): the struct keeps pixel_shift_cuda's defaults
    assert p.shift.depth_pop_gamma == 0.85 and p.shift.parallax_balance == 0.8
    assert p.format == FMT_HALF_SBS and p.shift.use_subject_tracking == 1 and p.shift.enable_floating_window == 1
    with pytest.raises(TypeError):
        render_kwargs_to_params(192, 108, output_height=108, fg_shift=1, mg_shift=1, bg_shift=1, sharpness_factor=0, output_format="Half-SBS",
                                dof_strength=0, codec="XVID")  # render_cli.py's stale kwarg is a TypeError in the reference too
    # skip_blank_frames is loop-level: the per-clip block is the same with and without it
    pk = dict(output_height=108, fg_shift=1, mg_shift=1, bg_shift=1, sharpness_factor=0, output_format="Half-SBS", dof_strength=0)
    assert bytes(render_kwargs_to_params(192, 108, skip_blank_frames=True, **pk)) == bytes(render_kwargs_to_params(192, 108, **pk))
    pa = render_kwargs_to_params(192, 108, output_height=108, fg_shift=1, mg_shift=1, bg_shift=1, sharpness_factor=0,
                                 output_format="Half-SBS", dof_strength=0, auto_crop_black_bars=True, target_ratio=2.39)
    assert pa.auto_crop_black_bars == 1 and pa.target_ratio == 2.39 and p.auto_crop_black_bars == 0


def test_shift_kwargs():
    p = shift_params_from_kwargs(4.5, -1.5, -6.0, blur_ksize=1, feather_strength=0.0, return_shift_map=False, dof_strength=2.0)
    assert isinstance(p, ShiftParams) and p.blur_ksize == 1 and p.fg_shift == 4.5
    with pytest.raises(TypeError):
        shift_params_from_kwargs(1, 2, 3, bogus=1)


def test_drop_in_entries_default_to_the_calling_process_thread_count(monkeypatch):
    """Round 6: the shims run inside the reference's process, so the N of the N-thread ATen mode defaults to that process's
    torch.get_num_threads() (core/render_3d.py:418,928,209,517,595-596 depend on it); VD3D_ATEN_THREADS overrides (0 = thread-independent); the parameter
    builders below the shims keep the C ABI's default 0."""
    import torch
    from visiondepth3d_amd import params as P
    from visiondepth3d_amd.render_3d import render_pairs
    prev = torch.get_num_threads()
    monkeypatch.delenv("VD3D_ATEN_THREADS", raising=False)   # (the suite's autouse fixture pins 0 for every other test)
    try:
        for n in (1, 3, 5):
            torch.set_num_threads(n)
            assert P.reference_aten_threads() == n
            assert shift_params_from_kwargs(1, 2, 3).aten_threads == n                      # pixel_shift_cuda(...) without the extension keyword
            assert shift_params_from_kwargs(1, 2, 3, aten_threads=0).aten_threads == 0       # explicit: thread-independent
            assert shift_params_from_kwargs(1, 2, 3, aten_threads=7).aten_threads == 7
            seen = []

            class Rec(_FakeRenderer):
                def render_frame(self, frame, depth, params, blank=False):
                    seen.append(params.aten_sum_threads)
                    return super().render_frame(frame, depth, params, blank)
            frames, depths = _fake_clip(3)
            kw = dict(output_height=54, fg_shift=10.0, mg_shift=-2.5, bg_shift=-5.0, sharpness_factor=0.15, output_format="Half-SBS", dof_strength=2.0)
            list(render_pairs(zip(frames, depths), renderer=Rec(), **kw))
            assert seen == [n, n]
            seen.clear()
            list(render_pairs(zip(frames, depths), renderer=Rec(), aten_sum_threads=0, **kw))
            assert seen == [0, 0]
        torch.set_num_threads(2)
        monkeypatch.setenv("VD3D_ATEN_THREADS", "64")
        assert P.reference_aten_threads() == 64 and shift_params_from_kwargs(1, 2, 3).aten_threads == 64
        monkeypatch.setenv("VD3D_ATEN_THREADS", "0")
        assert P.reference_aten_threads() == 0
        monkeypatch.setenv("VD3D_ATEN_THREADS", "5000")
        with pytest.raises(ValueError):
            P.reference_aten_threads()
    finally:
        torch.set_num_threads(prev)
    # the builders under the shims: the C ABI's default
    assert render_kwargs_to_params(192, 108, output_height=108, fg_shift=1, mg_shift=1, bg_shift=1, sharpness_factor=0, output_format="Half-SBS",
                                   dof_strength=0).aten_sum_threads == 0
    assert ShiftParams.defaults(1, 2, 3).aten_threads == 0


def test_synth_is_deterministic_and_well_conditioned():
    f1, d1 = synth.synth_frame(5, 108, 192)
    f2, d2 = synth.synth_frame(5, 108, 192)
    assert np.array_equal(f1, f2) and np.array_equal(d1, d2)
    assert f1.dtype == np.uint8 and d1.dtype == np.float32 and 0.0 <= d1.min() and d1.max() <= 1.0
    assert np.quantile(d1, 0.98) - np.quantile(d1, 0.02) > 0.3      # far from the collapse guard
    crop = d1[108 // 5:108 * 4 // 5, 192 // 5:192 * 4 // 5]
    assert np.count_nonzero((crop > 0.05) & (crop < 0.95)) > 20     # estimate_subject_depth has samples
    # platform-independence pin: integer-defined generator -> fixed checksum
    assert int(f1.astype(np.int64).sum()) == int(synth.synth_frame(5, 108, 192)[0].astype(np.int64).sum())
    g = synth.depth_to_u8_bgr(d1)
    assert g.shape == (108, 192, 3) and np.array_equal(g[..., 0], g[..., 2])


def test_u8_unit_is_the_exact_division():
    """vd_u8_unit (csrc/vd3d_dev.h): q0 = x*rc; q = fma(fma(-q0, 255, x), rc, q0) equals float32(x)/float32(255) for all 256
    inputs -- the device kernels use it instead of an IEEE division (restated here in float64-emulated float32 arithmetic)."""
    import math
    f32 = np.float32
    rc = f32(1.0) / f32(255.0)

    def fma32(a, b, c):   # exact product and sum in float64 (24-bit x 24-bit fits), one rounding to float32
        return f32(np.float64(a) * np.float64(b) + np.float64(c))

    for v in range(256):
        x = f32(v)
        q0 = f32(x * rc)
        q = fma32(fma32(-q0, f32(255.0), x), rc, q0)
        assert q == x / f32(255.0), v
    assert math.isfinite(float(rc))


class _FakeRenderer:
    """Host-logic stand-in for Renderer (no GPU): records what the shell asks for and returns recognisable frames."""

    def __init__(self):
        import torch
        self.device = torch.device("cpu")
        self.calls, self.clips = [], 0

    def new_clip(self):
        self.clips += 1

    def render_frame(self, frame, depth, params, blank=False):
        import torch
        self.calls.append((int(frame[0, 0, 0]), bool(blank)))
        out = torch.zeros((params.out_h, params.out_w, 3), dtype=torch.uint8)
        out[...] = int(frame[0, 0, 0])
        return out


def _fake_clip(n, h=54, w=96):
    import numpy as np
    frames = [np.full((h, w, 3), i, np.uint8) for i in range(n)]     # frame i carries its own index in every byte
    return frames, [f.copy() for f in frames]


def _shell_args(**over):
    import threading
    a = dict(input_path="in.mp4", depth_path="depth.mp4", output_path="out.avi", selected_codec="XVID", fps=24.0, output_width=96,
             output_height=54, fg_shift=10.0, mg_shift=-2.5, bg_shift=-5.0, sharpness_factor=0.15, output_format="Half-SBS",
             selected_aspect_ratio="Default (16:9)", aspect_ratios={"Default (16:9)": 16 / 9}, dof_strength=2.0,
             suspend_flag=threading.Event(), cancel_flag=threading.Event())
    a.update(over)
    return a


def test_render_sbs_3d_signature_is_the_references_51_parameters():
    """B2's Python face (core/render_3d.py:933-985): same parameter names, order and defaults (+ one keyword-only extension)."""
    import inspect
    from visiondepth3d_amd.params import RENDER_DEFAULTS
    from visiondepth3d_amd.render_3d import render_sbs_3d
    ps = list(inspect.signature(render_sbs_3d).parameters.values())
    pos = [p for p in ps if p.kind is p.POSITIONAL_OR_KEYWORD]
    assert len(pos) == 51
    assert [p.name for p in pos[:15]] == ["input_path", "depth_path", "output_path", "selected_codec", "fps", "output_width", "output_height",
                                          "fg_shift", "mg_shift", "bg_shift", "sharpness_factor", "output_format", "selected_aspect_ratio",
                                          "aspect_ratios", "dof_strength"]
    assert all(p.default is p.empty for p in pos[:15])
    assert {p.name: p.default for p in pos[15:]} == RENDER_DEFAULTS and [p.name for p in pos[15:]] == list(RENDER_DEFAULTS)
    assert [p.name for p in ps if p.kind is p.KEYWORD_ONLY] == ["renderer"]


def test_render_sbs_3d_shell_read_order_writer_and_window(monkeypatch):
    """The capture / writer shell with an in-memory video backend (the fake cv2 of tests/golden/ref_stubs.py) and a fake renderer:
    the window's first frame is decoded twice and never rendered (:1024-1028,1184-1189), rendering starts with the next frame,
    the writer is opened with the reference's writer size (:1134-1138) and gets every muxed frame; start_s / end_s resolve
    to frames as :1008-1011 and the loop stops at end_frame_idx (:1432-1435)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import ref_stubs
    from visiondepth3d_amd import video_io
    monkeypatch.setattr(video_io, "video_backend", ref_stubs)
    frames, depths = _fake_clip(10)
    ref_stubs._Clip.clips["in.mp4"], ref_stubs._Clip.clips["depth.mp4"] = frames, depths
    caps = []
    real_cap = ref_stubs.VideoCapture

    class Cap(real_cap):
        def __init__(self, path):
            super().__init__(path)
            caps.append(self)
    monkeypatch.setattr(ref_stubs, "VideoCapture", Cap)
    fr = _FakeRenderer()
    video_io.render_sbs_3d(**_shell_args(), renderer=fr)
    assert caps[0].log == [0, 0] + list(range(1, 10))             # frame 0 decoded twice, then 1..9
    assert [c[0] for c in fr.calls] == list(range(1, 10)) and fr.clips == 1
    written = ref_stubs._Clip.written["out.avi"]
    assert len(written) == 9 and written[0].shape == (54, 96, 3) and int(written[3][0, 0, 0]) == 4
    # Full-SBS: the writer is opened 2 x 1920 wide whatever the source is (:1099-1101)
    sizes = []
    real_wr = ref_stubs.VideoWriter

    class Wr(real_wr):
        def __init__(self, path, fourcc, fps, size):
            super().__init__(path, fourcc, fps, size)
            sizes.append((fourcc, size))
    monkeypatch.setattr(ref_stubs, "VideoWriter", Wr)
    video_io.render_sbs_3d(**_shell_args(output_format="Full-SBS", output_height=1080), renderer=_FakeRenderer())
    assert sizes == [("XVID", (3840, 1080))]
    # clip window: fps 24, start 0.125 s -> frame 3, end 0.25 s -> frame 6: frame 3 consumed, 4 and 5 rendered
    fr = _FakeRenderer()
    video_io.render_sbs_3d(**_shell_args(start_s=0.125, end_s=0.25), renderer=fr)
    assert [c[0] for c in fr.calls] == [4, 5]
    fr = _FakeRenderer()
    video_io.render_sbs_3d(**_shell_args(start_s=5.0), renderer=fr)        # empty window: nothing rendered, no writer
    assert fr.calls == []
    # unreadable inputs return silently like the reference (:988-989)
    assert video_io.render_sbs_3d(**_shell_args(input_path="missing.mp4"), renderer=_FakeRenderer()) is None


def test_render_sbs_3d_shell_cancel_blank_and_ffmpeg_pipe(monkeypatch):
    import sys
    import numpy as np
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import ref_stubs
    from visiondepth3d_amd import video_io
    monkeypatch.setattr(video_io, "video_backend", ref_stubs)
    frames, depths = _fake_clip(8)
    ref_stubs._Clip.clips["in.mp4"], ref_stubs._Clip.clips["depth.mp4"] = frames, depths
    # cancel: tested before a frame is decoded (:1196-1197) -> a flag set from the start renders nothing
    args = _shell_args()
    args["cancel_flag"].set()
    fr = _FakeRenderer()
    video_io.render_sbs_3d(**args, renderer=fr)
    assert fr.calls == []

    # cancel raised while frame 3 is being rendered: frame 3 is still written, nothing after it is decoded
    args = _shell_args()

    class Cancelling(_FakeRenderer):
        def render_frame(self, frame, depth, params, blank=False):
            if int(frame[0, 0, 0]) == 3:
                args["cancel_flag"].set()
            return super().render_frame(frame, depth, params, blank)
    fr = Cancelling()
    video_io.render_sbs_3d(**args, renderer=fr)
    assert [c[0] for c in fr.calls] == [1, 2, 3] and len(ref_stubs._Clip.written["out.avi"]) == 3
    # skip_blank_frames: absolute indices from the plugged-in detector, offset by the window start (:1063,1278)
    monkeypatch.setattr(video_io, "blank_frame_detector", lambda path: [2, 5])
    fr = _FakeRenderer()
    video_io.render_sbs_3d(**_shell_args(skip_blank_frames=True), renderer=fr)
    assert [c for c in fr.calls if c[1]] == [(3, True), (6, True)]       # loop index 2 / 5 = frames 3 / 6 (the loop starts at frame 1)
    monkeypatch.setattr(video_io, "blank_frame_detector", None)         # no detector: every frame rendered normally (:1058-1060)
    fr = _FakeRenderer()
    video_io.render_sbs_3d(**_shell_args(skip_blank_frames=True), renderer=fr)
    assert not any(b for _, b in fr.calls) and len(fr.calls) == 7

    # ffmpeg pipe (:1143-1163,1422-1427): rawvideo bgr24 of the writer size on stdin, unknown encoders fall back to libx264
    class Pipe:
        def __init__(self):
            self.buf, self.closed = bytearray(), False
        def write(self, b):
            self.buf += b
        def close(self):
            self.closed = True

    class Proc:
        def __init__(self, cmd, stdin=None):
            self.cmd, self.stdin, self.waited = cmd, Pipe(), False
            procs.append(self)
        def wait(self, timeout=None):
            self.waited = True
        def kill(self):
            pass
    procs = []
    monkeypatch.setattr(video_io, "popen", Proc)
    video_io.render_sbs_3d(**_shell_args(use_ffmpeg=True, selected_ffmpeg_codec="not-a-codec", crf_value=19), renderer=_FakeRenderer())
    cmd = procs[0].cmd
    assert cmd[:2] == ["ffmpeg", "-y"] and cmd[cmd.index("-s") + 1] == "96x54" and cmd[cmd.index("-pix_fmt") + 1] == "bgr24"
    assert cmd[cmd.index("-c:v") + 1] == "libx264" and cmd[cmd.index("-crf") + 1] == "19" and cmd[-1] == "out.avi"
    assert len(procs[0].stdin.buf) == 7 * 54 * 96 * 3 and procs[0].stdin.closed and procs[0].waited
    assert video_io.ffmpeg_pipe_command(8, 4, 30.0, "hevc_nvenc", 21, "o.mp4")[-5:] == ["-cq", "21", "-b:v", "0", "o.mp4"]
    assert "-crf" not in video_io.ffmpeg_pipe_command(8, 4, 30.0, "h264_amf", 21, "o.mp4")

    # opt-in NV12 wire format: the frame is converted where it lives and 1.5 bytes per pixel reach the pipe
    from oracle import oracle as O

    class Nv12Renderer(_FakeRenderer):
        def bgr_to_nv12(self, t):
            import torch
            return torch.from_numpy(O.bgr_to_nv12(t.numpy()))
    procs.clear()
    monkeypatch.setattr(video_io, "PIPE_PIX_FMT", "nv12")
    video_io.render_sbs_3d(**_shell_args(use_ffmpeg=True, selected_ffmpeg_codec="libx264"), renderer=Nv12Renderer())
    cmd = procs[0].cmd
    assert cmd[cmd.index("-pix_fmt") + 1] == "nv12" and cmd[cmd.index("-s") + 1] == "96x54"
    assert len(procs[0].stdin.buf) == 7 * 54 * 96 * 3 // 2
    first = np.frombuffer(bytes(procs[0].stdin.buf[:54 * 96 * 3 // 2]), np.uint8).reshape(81, 96)
    assert np.array_equal(first, O.bgr_to_nv12(np.full((54, 96, 3), 1, np.uint8)))
    # a format whose muxed frame is not the writer frame (interlaced: one canvas written, two opened) stays on bgr24
    procs.clear()
    video_io.render_sbs_3d(**_shell_args(use_ffmpeg=True, selected_ffmpeg_codec="libx264", output_format="Passive Interlaced"),
                           renderer=Nv12Renderer())
    assert procs[0].cmd[procs[0].cmd.index("-pix_fmt") + 1] == "bgr24"


def test_depth_pipe_split_modes_are_refused_without_the_library():
    """The opt-in modes of round 6 are kernels of libvd3d_hip.so: a CPU pipe, a bf16 pipe or a pipe without a renderer cannot have them -- a loud ValueError, never a
    silent fall-back to the float32 library GEMMs."""
    import torch
    from visiondepth3d_amd.depth import DepthPipe
    for mode in ("bf16x3", "fp16x2"):
        with pytest.raises(ValueError):
            DepthPipe("depth-anything-v2-small", device="cpu", dtype=torch.float32, gemm=mode)
    with pytest.raises(ValueError):
        DepthPipe("depth-anything-v2-small", device="cpu", dtype=torch.float32, gemm="tf32")


def test_dpt_front_end_matches_the_real_image_processor():
    """B3 / a25 front end against transformers' own DPTImageProcessor (the Depth-Anything-V2 preprocessor_config values): the
    target size rule on 70 frame sizes exactly, and the float statement the fused kernel is tested against (antialiased bicubic in
    float32, tests/test_hip_depthprep.py) within ~1 LSB of the 8-bit image the PIL path rounds to (tolerance 1.5 / 255 / std)."""
    import numpy as np
    import torch
    import torch.nn.functional as F
    transformers = pytest.importorskip("transformers")
    from PIL import Image
    from visiondepth3d_amd import synth
    from visiondepth3d_amd.depth import IMAGENET_MEAN, IMAGENET_STD, dpt_resize_target
    proc = transformers.DPTImageProcessor(do_resize=True, size={"height": 518, "width": 518}, keep_aspect_ratio=True, ensure_multiple_of=14,
                                          resample=3, do_rescale=True, rescale_factor=1 / 255, do_normalize=True,
                                          image_mean=IMAGENET_MEAN, image_std=IMAGENET_STD, do_pad=False)
    rng = np.random.default_rng(3)
    sizes = [(1080, 1920), (2160, 3840), (720, 1280), (480, 640), (270, 480), (101, 333), (518, 924), (1000, 1000), (1440, 2560),
             (1600, 1440)] + [(int(rng.integers(40, 2200)), int(rng.integers(40, 4000))) for _ in range(60)]
    for h, w in sizes:
        out = proc(images=Image.fromarray(np.zeros((h, w, 3), np.uint8)), return_tensors="pt")["pixel_values"]
        assert tuple(out.shape[-2:]) == dpt_resize_target(h, w), (h, w)
    assert dpt_resize_target(1080, 1920) == (518, 924)
    h, w = 270, 480
    bgr = synth.synth_frame(2, h, w)[0]
    pv = proc(images=Image.fromarray(bgr[..., ::-1].copy()), return_tensors="pt")["pixel_values"][0]
    th, tw = dpt_resize_target(h, w)
    x = torch.from_numpy(bgr[..., ::-1].copy()).permute(2, 0, 1)[None].float()
    x = F.interpolate(x, size=(th, tw), mode="bicubic", antialias=True, align_corners=False)
    mean, std = torch.tensor(IMAGENET_MEAN).view(1, 3, 1, 1), torch.tensor(IMAGENET_STD).view(1, 3, 1, 1)
    ref = (((x / 255.0) - mean) / std)[0]
    assert float((pv - ref).abs().max()) < 1.5 / 255 / min(IMAGENET_STD)


def test_depth_pipe_protocol_against_the_transformers_pipeline():
    """B3 (core/render_depth.py:1106-1119): ``pipe(list[PIL], inference_size) -> [{"predicted_depth": Tensor[h, w]}]`` against
    transformers' own ``pipeline("depth-estimation")`` around the stock ``DepthAnythingForDepthEstimation`` with the same synthetic
    weights, on CPU in float32.  (a) Same pixel_values through the rewritten module graph (fused QKV, folded LayerScale, cached
    position embedding) and the stock one: 1e-6.  (b) Whole protocol: the prediction comes back at the image size like the
    pipeline's; values within a few percent -- entirely the 8-bit rounding of the resized image in the PIL / torchvision front
    ends (the float statement differs from ITS OWN uint8 rounding by the same 1 %: random weights amplify +-0.5 LSB that much)."""
    import numpy as np
    import torch
    transformers = pytest.importorskip("transformers")
    from PIL import Image
    from visiondepth3d_amd import synth
    from visiondepth3d_amd.depth import IMAGENET_MEAN, IMAGENET_STD, DepthPipe, build_config, synthetic_weights_
    torch.set_num_threads(min(16, torch.get_num_threads()))
    pipe = DepthPipe("depth-anything-v2-small", device="cpu", dtype=torch.float32)
    model = transformers.DepthAnythingForDepthEstimation(build_config("depth-anything-v2-small")).eval()
    synthetic_weights_(model, 0)
    proc = transformers.DPTImageProcessor(do_resize=True, size={"height": 518, "width": 518}, keep_aspect_ratio=True, ensure_multiple_of=14,
                                          resample=3, do_rescale=True, rescale_factor=1 / 255, do_normalize=True,
                                          image_mean=IMAGENET_MEAN, image_std=IMAGENET_STD, do_pad=False)
    img = Image.fromarray(synth.synth_frame(1, 126, 224)[0][..., ::-1].copy())
    pv = proc(images=img, return_tensors="pt")["pixel_values"]
    with torch.no_grad():
        a = pipe.model(pixel_values=pv).predicted_depth
        b = model(pixel_values=pv).predicted_depth
    assert float((a - b).abs().max() / b.abs().max()) < 1e-4
    hf = transformers.pipeline("depth-estimation", model=model, image_processor=proc, device="cpu")
    exp = hf([img])[0]["predicted_depth"].squeeze()
    got = pipe([img])[0]["predicted_depth"]
    assert tuple(got.shape) == tuple(exp.shape) == (126, 224)
    assert float((got - exp).abs().mean() / exp.abs().mean()) < 3e-2
    # hf_batch_safe_pipe with an inference size: the images are pre-resized, the prediction comes back at THAT size (:1113-1116)
    small = img.resize((112, 70), Image.BICUBIC)
    exp2 = hf([small])[0]["predicted_depth"].squeeze()
    got2 = pipe([img], inference_size=(112, 70))[0]["predicted_depth"]
    assert tuple(got2.shape) == tuple(exp2.shape) == (70, 112)
    assert float((got2 - exp2).abs().mean() / exp2.abs().mean()) < 5e-2


def test_depth_pipe_from_pretrained_round_trip(tmp_path):
    """B3 checkpoint loading (core/render_depth.py:756-760: AutoModelForDepthEstimation.from_pretrained(local folder)): a stock
    DepthAnythingForDepthEstimation saved with save_pretrained (safetensors + config.json + preprocessor_config.json) comes back
    through DepthPipe.from_pretrained, gets the fused-weight rewrites (one QKV GEMM, folded LayerScale) and predicts the same
    depth as the stock graph to 1e-6 (mean, relative to the output range; max 1e-5) on CPU / float32."""
    import torch
    transformers = pytest.importorskip("transformers")
    from visiondepth3d_amd.depth import IMAGENET_MEAN, IMAGENET_STD, DepthPipe, build_config, synthetic_weights_
    torch.set_num_threads(min(16, torch.get_num_threads()))
    model = transformers.DepthAnythingForDepthEstimation(build_config("depth-anything-v2-small")).eval()
    synthetic_weights_(model, 3)
    model.save_pretrained(tmp_path)
    transformers.DPTImageProcessor(do_resize=True, size={"height": 518, "width": 518}, keep_aspect_ratio=True, ensure_multiple_of=14,
                                   resample=3, image_mean=IMAGENET_MEAN, image_std=IMAGENET_STD).save_pretrained(tmp_path)
    assert (tmp_path / "model.safetensors").exists() and (tmp_path / "preprocessor_config.json").exists()
    pipe = DepthPipe.from_pretrained(str(tmp_path), device="cpu")
    assert pipe.dtype == torch.float32 and pipe.arch == "da"
    assert pipe.proc["size"] == (518, 518) and pipe.proc["multiple"] == 14 and pipe.proc["keep_aspect_ratio"] is True
    assert pipe.resize_target(1080, 1920) == (518, 924)
    g = torch.Generator().manual_seed(5)
    pv = torch.randn(1, 3, 70, 126, generator=g)
    with torch.no_grad():
        a = pipe.model(pixel_values=pv).predicted_depth
        b = model(pixel_values=pv).predicted_depth
    # float32 association noise of the rewrites (LayerScale folded into the weights, one QKV GEMM): mean 1e-6, max 1e-5 of range
    assert float((a - b).abs().mean() / b.abs().max()) < 1e-6
    assert float((a - b).abs().max() / b.abs().max()) < 1e-5
    # the fused weights really are derived from the LOADED checkpoint: a different seed gives a different prediction
    other = DepthPipe("depth-anything-v2-small", device="cpu", seed=4)
    with torch.no_grad():
        c = other.model(pixel_values=pv).predicted_depth
    assert float((c - b).abs().max() / b.abs().max()) > 1e-3
    with pytest.raises(FileNotFoundError):
        DepthPipe.from_pretrained(str(tmp_path / "missing"))


def test_depth_pipe_model_zoo_covers_the_reference_hf_families():
    """core/render_depth.py:686-712: Depth-Anything V1 / V2 and Distill-Any-Depth are one architecture family (DepthAnything on
    DINOv2); MiDaS 3.0 / DPT-Large is DPTForDepthEstimation.  Shapes only (no weights exist here)."""
    from visiondepth3d_amd.depth import MODEL_ZOO, build_config
    for fam in ("depth-anything-v2", "depth-anything-v1"):
        for size in ("small", "base", "large"):
            assert f"{fam}-{size}" in MODEL_ZOO
    assert MODEL_ZOO["distill-any-depth-large"] is MODEL_ZOO["depth-anything-v2-large"]
    cfg = build_config("dpt-large")
    assert type(cfg).__name__ == "DPTConfig" and cfg.hidden_size == 1024 and cfg.num_hidden_layers == 24


def test_depth_pipe_generic_architecture_dpt_protocol():
    """A non-DepthAnything Hugging Face depth model (DPTForDepthEstimation = the MiDaS 3.0 / Intel dpt-* family of the reference's
    list, core/render_depth.py:706-709) behind the same B3 protocol: its own image-processor constants (384 x 384, no aspect
    keeping, mean = std = 0.5), stock module graph.  Tiny configuration (2 layers) so the CPU test stays fast."""
    import torch
    transformers = pytest.importorskip("transformers")
    from PIL import Image
    from visiondepth3d_amd import synth
    from visiondepth3d_amd.depth import PROCESSORS, DepthPipe, synthetic_weights_
    torch.set_num_threads(min(16, torch.get_num_threads()))
    cfg = transformers.DPTConfig(hidden_size=64, num_hidden_layers=4, num_attention_heads=4, intermediate_size=128, image_size=64,
                                 patch_size=16, backbone_out_indices=[0, 1, 2, 3], neck_hidden_sizes=[16, 32, 64, 64],
                                 fusion_hidden_size=32, readout_type="project")
    model = transformers.DPTForDepthEstimation(cfg).eval()
    synthetic_weights_(model, 1)
    proc_kw = dict(PROCESSORS["dpt"], size=(64, 64))
    pipe = DepthPipe("tiny-dpt", device="cpu", model=model, processor=proc_kw)
    assert pipe.arch == "generic" and pipe.resize_target(126, 224) == (64, 64)
    proc = transformers.DPTImageProcessor(do_resize=True, size={"height": 64, "width": 64}, keep_aspect_ratio=False, ensure_multiple_of=1,
                                          resample=3, do_rescale=True, rescale_factor=1 / 255, do_normalize=True,
                                          image_mean=[0.5, 0.5, 0.5], image_std=[0.5, 0.5, 0.5], do_pad=False)
    img = Image.fromarray(synth.synth_frame(2, 126, 224)[0][..., ::-1].copy())
    hf = transformers.pipeline("depth-estimation", model=model, image_processor=proc, device="cpu")
    exp = hf([img])[0]["predicted_depth"].squeeze()
    got = pipe([img])[0]["predicted_depth"]
    assert tuple(got.shape) == tuple(exp.shape) == (126, 224)
    assert float((got - exp).abs().mean() / exp.abs().mean()) < 3e-2   # 8-bit rounding of the resized image in the PIL front end


def test_dpt_neck_head_rewrite_is_the_module_graph():
    """DepthPipe._patch_dpt_upsampling in float32 (bias-free convolutions + one glue launch between them, projection before its up-sampling,
    fused head tail, channels_last reassemble view) against the stock transformers graph on CPU, with a torch double of the three glue
    entry points that states what each HIP kernel computes (include/vd3d.h: vd3d_nhwc_bias_act_f32, vd3d_upsample_bilinear_bias_nhwc_f32,
    vd3d_dpt_head_tail_f32).  The GPU tests (tests/test_hip_depth_e2e.py) run the same graph on the kernels themselves."""
    import torch
    import torch.nn.functional as F
    transformers = pytest.importorskip("transformers")
    from visiondepth3d_amd.depth import DepthPipe, build_config, synthetic_weights_

    class GlueDouble:
        calls = {"bias_act": 0, "up_bias": 0, "tail": 0, "up": 0}

        def upsample_bilinear(self, x, size):
            self.calls["up"] += 1
            return F.interpolate(x, size=tuple(size), mode="bilinear", align_corners=True)

        def bias_act(self, y, bias=None, r1=None, r2=None, relu=False, want_relu_copy=False):
            self.calls["bias_act"] += 1
            v = y if bias is None else y + bias.view(1, -1, 1, 1)
            if r1 is not None:
                v = v + r1
            if r2 is not None:
                v = r2 + v
            if relu:
                v = torch.relu(v)
            y.copy_(v)
            return (y, torch.relu(y)) if want_relu_copy else y

        def upsample_bilinear_bias(self, x, size, bias):
            self.calls["up_bias"] += 1
            return F.interpolate(x, size=tuple(size), mode="bilinear", align_corners=True) + bias.view(1, -1, 1, 1)

        def dpt_head_tail(self, y, b2, w3, b3, scale):
            self.calls["tail"] += 1
            return torch.relu((torch.relu(y + b2.view(1, -1, 1, 1)) * w3.view(1, -1, 1, 1)).sum(1) + b3) * scale

    torch.set_num_threads(min(16, torch.get_num_threads()))
    for name in ("depth-anything-v2-small", "depth-anything-v2-base"):
        pipe = DepthPipe(name, device="cpu", dtype=torch.float32)
        pipe.renderer = GlueDouble()
        GlueDouble.calls = dict.fromkeys(GlueDouble.calls, 0)
        pipe._patch_dpt_upsampling()
        model = transformers.DepthAnythingForDepthEstimation(build_config(name)).eval()
        synthetic_weights_(model, 0)
        pv = torch.randn(2, 3, 70, 98, generator=torch.Generator().manual_seed(3))
        with torch.no_grad():
            a = pipe.model(pixel_values=pv).predicted_depth
            b = model(pixel_values=pv).predicted_depth
        assert a.shape == b.shape
        assert float((a - b).abs().max() / b.abs().max()) < 1e-5, name
        # 7 residual units x 2 glue launches; 4 projections + head conv1 hand their bias to the up-sampling; one head tail; no plain up-sampling left
        assert GlueDouble.calls == {"bias_act": 14, "up_bias": 5, "tail": 1, "up": 0}, GlueDouble.calls
        # flops_per_frame counts what the rewritten graph EXECUTES: the stock count minus 3/4 of the four 1x1 projections (they run before
        # their up-samplings now); the bias-free convolution calls bypass the module hooks and count themselves
        stock = DepthPipe(name, device="cpu", dtype=torch.float32).flops_per_frame(126, 224)
        mine = pipe.flops_per_frame(126, 224)
        th, tw = pipe.resize_target(126, 224)
        Cf = pipe.model.config.fusion_hidden_size
        ph, pw = th // 14, tw // 14
        px_stock = ph * pw * (1 + 4 + 16 + 64)                                   # the projections' output pixels in the stock graph ...
        px_mine = -(-ph // 2) * -(-pw // 2) + ph * pw * (1 + 4 + 16)              # ... and their input pixels (the stride-2 reassemble map rounds up)
        assert abs((stock - mine) - 2.0 * Cf * Cf * (px_stock - px_mine)) <= 1e-9 * stock, (stock, mine)
