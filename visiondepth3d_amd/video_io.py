"""``render_sbs_3d`` -- the I/O shell around the per-frame path, with the reference's explicit 51-parameter signature
(core/render_3d.py:933-985) so that the GUI's ``process_video`` call (:1701-1753) and ``render_cli.py`` work unchanged.

What the shell does (all per-frame work is ``render_3d.render_clip`` -> ``vd3d_render_frame``):
  * opens both videos through the video backend (``cv2`` unless ``video_backend`` is set -- tests inject an in-memory one),
    resolves the ``start_s`` / ``end_s`` clip window in frames, and reproduces the reference's read order: the frame at the
    window start is decoded twice (:1024-1028 and :1184-1189) and never rendered; rendering starts with the frame after it;
  * opens the writer with the reference's writer size (:1134-1138) -- ``cv2.VideoWriter`` or the
    ``ffmpeg -f rawvideo -pix_fmt bgr24`` stdin pipe of :1143-1163, fed with ``final.tobytes()`` (:1422-1427);
  * honours ``cancel_flag`` / ``suspend_flag`` (threading.Event-like) and drives ``progress`` / ``progress_label``.
``skip_blank_frames`` needs the list of blank frames; the reference asks ffmpeg's blackdetect filter
(core/ffmpeg_blackdetect.py, OUT OF SCOPE here).  A caller that has that function plugs it in as
``blank_frame_detector``; without one the shell renders every frame, which is what the reference does when its detection
fails (:1058-1060).
"""
from __future__ import annotations

import subprocess
import time

import numpy as np

video_backend = None          # module-like object with VideoCapture / VideoWriter / VideoWriter_fourcc / CAP_PROP_*; None -> cv2
blank_frame_detector = None   # callable(input_path) -> iterable of absolute blank-frame indices, or None
popen = subprocess.Popen      # the ffmpeg pipe is opened through this (tests replace it)

# encoder names the reference accepts for its ffmpeg pipe (values of its codec table, core/render_3d.py:49-75); anything else
# falls back to libx264 (:1037-1044)
_SW = ("libx264", "libx265", "libaom-av1", "libsvtav1", "mp4v", "XVID", "DIVX")
_HW = tuple(f"{c}_{v}" for v in ("nvenc", "amf", "qsv") for c in ("h264", "hevc", "av1"))
KNOWN_FFMPEG_ENCODERS = frozenset(_SW + _HW)


def _backend():
    if video_backend is not None:
        return video_backend
    try:
        import cv2
    except ImportError as e:
        raise ImportError("render_sbs_3d needs OpenCV for VideoCapture / VideoWriter (or set visiondepth3d_amd.video_io."
                          "video_backend); render_clip() renders in-memory frames without it") from e
    return cv2


# Wire format of the encoder pipe: "bgr24" is the reference's (:1146); "nv12" (opt-in) converts on the device and halves the bytes that
# cross PCIe and the pipe -- ffmpeg then skips its own RGB -> YUV pass for the yuv420p output.
RENDER_BATCH = 8     # frames per renderer step (render_3d.render_pairs): 1 = one vd3d_render_frame per frame; results are identical
PIPE_PIX_FMT = "bgr24"


def ffmpeg_pipe_command(out_w, out_h, fps, encoder, crf_value, output_path, pix_fmt="bgr24"):
    """The rawvideo-over-stdin command line of :1143-1161 (bgr24 frames of the writer size in, yuv420p out)."""
    if not isinstance(encoder, str) or not encoder.strip() or encoder not in KNOWN_FFMPEG_ENCODERS:
        encoder = "libx264"
    cmd = ["ffmpeg", "-y", "-f", "rawvideo", "-vcodec", "rawvideo", "-pix_fmt", pix_fmt, "-s", f"{out_w}x{out_h}", "-r", str(fps),
           "-i", "-", "-an", "-c:v", encoder, "-preset", "slow", "-pix_fmt", "yuv420p"]
    if encoder.startswith("libx"):
        cmd += ["-crf", str(crf_value)]
    elif "nvenc" in encoder:
        cmd += ["-cq", str(crf_value), "-b:v", "0"]
    return cmd + [output_path]


def clip_window(total_frames, fps, start_s, end_s):
    """(start_frame, end_frame, frames_in_window) of :992-1017, or None when the window is empty."""
    dur_ms = total_frames / max(fps, 1e-6) * 1000.0
    lo = max(0.0, (start_s or 0.0) * 1000.0)
    hi = dur_ms if end_s is None else min(dur_ms, end_s * 1000.0)
    if lo >= hi - 0.5:
        return None
    a, b = int(round(lo / 1000.0 * fps)), int(round(hi / 1000.0 * fps))
    return a, b, max(0, b - a)


class _Sink:
    """One of the reference's two writers behind one write()/close() pair."""

    def __init__(self, cv, output_path, size, fps, use_ffmpeg, fourcc, encoder, crf_value, pix_fmt="bgr24", renderer=None, frame_size=None):
        self.proc = self.writer = None
        # NV12 only when the muxed frame IS the writer frame (not the interlaced / anaglyph mismatch of :851-858) and 4:2:0 divides it
        self.nv12 = (bool(use_ffmpeg) and pix_fmt == "nv12" and size[0] % 2 == 0 and size[1] % 2 == 0
                     and (frame_size is None or tuple(frame_size) == tuple(size)))
        self.renderer = renderer
        if use_ffmpeg:
            self.proc = popen(ffmpeg_pipe_command(size[0], size[1], fps, encoder, crf_value, output_path,
                                                  "nv12" if self.nv12 else "bgr24"), stdin=subprocess.PIPE)
        else:
            self.writer = cv.VideoWriter(output_path, cv.VideoWriter_fourcc(*fourcc), fps, size)
            if not self.writer.isOpened():
                self.writer = None
                raise OSError("VideoWriter failed to open (codec / fourcc / path)")

    def write(self, frame) -> bool:
        if self.proc is not None:
            try:
                if self.nv12:   # colour conversion on the frame that is still in HBM; 1.5 bytes per pixel come back
                    import torch
                    from .render_3d import default_renderer
                    r = self.renderer or default_renderer()
                    t = frame if torch.is_tensor(frame) else torch.from_numpy(np.ascontiguousarray(frame, dtype=np.uint8))
                    nv = r.bgr_to_nv12(t)
                    frame = (r.to_host(nv) if hasattr(r, "to_host") else nv.cpu()).numpy()   # to_host: ordered behind a private stream
                self.proc.stdin.write(np.ascontiguousarray(frame, dtype=np.uint8).tobytes())
            except Exception as e:   # a dead encoder ends the render like in the reference (:1424-1426)
                print(f"FFmpeg write error: {e}")
                return False
        else:
            self.writer.write(frame)
        return True

    def close(self, cancelled: bool):
        if self.proc is not None:
            for fn in ((lambda: self.proc.stdin.close()), (lambda: self.proc.kill() if cancelled else self.proc.wait(timeout=5))):
                try:
                    fn()
                except Exception:
                    pass
        elif self.writer is not None:
            self.writer.release()


def render_sbs_3d(
    input_path,
    depth_path,
    output_path,
    selected_codec,
    fps,
    output_width,
    output_height,
    fg_shift,
    mg_shift,
    bg_shift,
    sharpness_factor,
    output_format,
    selected_aspect_ratio,
    aspect_ratios,
    dof_strength,
    feather_strength=0.0,
    blur_ksize=1,
    use_ffmpeg=False,
    selected_ffmpeg_codec=None,
    crf_value=23,
    use_subject_tracking=False,
    use_floating_window=False,
    max_pixel_shift_percent=0.02,
    progress=None,
    progress_label=None,
    suspend_flag=None,
    cancel_flag=None,
    auto_crop_black_bars=False,
    parallax_balance=0.8,
    preserve_original_aspect=False,
    zero_parallax_strength=0.0,
    enable_edge_masking=True,
    enable_feathering=True,
    skip_blank_frames=False,
    original_video_width=None,
    original_video_height=None,
    convergence_strength=0.0,
    enable_dynamic_convergence=True,
    ipd_factor=1.0,
    depth_pop_gamma=0.85,
    depth_pop_mid=0.50,
    depth_stretch_lo=0.05,
    depth_stretch_hi=0.95,
    fg_pop_multiplier=1.20,
    bg_push_multiplier=1.10,
    subject_lock_strength=1.00,
    color_saturation=1.0,
    color_contrast=1.0,
    color_brightness=0.0,
    start_s=None,
    end_s=None,
    *,
    renderer=None,
):
    """Reference signature and behaviour (core/render_3d.py:933-1500); ``output_width`` is accepted and unused exactly like
    there (the loop derives every width from ``output_height`` and the aspect ratio, :1086-1138).  ``renderer`` (keyword-only
    extension): a ``Renderer`` to use instead of the module-level default."""
    from .geometry import plan_geometry
    from .render_3d import render_pairs
    cv = _backend()
    cap, dcap = cv.VideoCapture(input_path), cv.VideoCapture(depth_path)
    if not cap.isOpened() or not dcap.isOpened():
        return
    sink, clip, cancelled = None, None, (lambda: cancel_flag is not None and cancel_flag.is_set())
    try:
        n_total = int(cap.get(cv.CAP_PROP_FRAME_COUNT))
        fps = cap.get(cv.CAP_PROP_FPS) or fps or 30.0
        win = clip_window(n_total, fps, start_s, end_s)
        if win is None:
            print("Invalid clip window; nothing to render.")
            return
        first_idx, end_idx, n_win = win
        probe = None
        for _ in range(2):   # the reference decodes the window's first frame twice and renders neither copy
            cap.set(cv.CAP_PROP_POS_FRAMES, first_idx)
            dcap.set(cv.CAP_PROP_POS_FRAMES, first_idx)
            ok_f, probe = cap.read()
            ok_d, _d = dcap.read()
            if not ok_f or not ok_d:
                return
        blank = None
        if skip_blank_frames and blank_frame_detector is not None:
            try:
                blank = set(blank_frame_detector(input_path))
            except Exception as e:
                print(f"Blank frame detection failed: {e}")
        label = selected_aspect_ratio.get() if hasattr(selected_aspect_ratio, "get") else selected_aspect_ratio
        target_ratio = aspect_ratios.get(label, 16 / 9)
        src_h, src_w = int(probe.shape[0]), int(probe.shape[1])
        if preserve_original_aspect and auto_crop_black_bars and (original_video_width is None or original_video_height is None):
            # the reference's fallback size is the first frame AFTER crop_black_bars_torch (core/render_3d.py:1066-1089)
            import torch
            from .render_3d import default_renderer
            top, bottom = (renderer or default_renderer()).detect_black_bars(torch.from_numpy(np.ascontiguousarray(probe)))
            original_video_width = src_w
            original_video_height = src_h - top - bottom if top + bottom < src_h else src_h   # :326 guard: an all-bar frame stays uncropped
        geom = plan_geometry(src_w, src_h, output_height, output_format, target_ratio, preserve_original_aspect,
                             original_video_width, original_video_height)
        try:
            sink = _Sink(cv, output_path, (geom["writer_w"], geom["writer_h"]), fps, use_ffmpeg, selected_codec,
                         selected_ffmpeg_codec, crf_value, pix_fmt=PIPE_PIX_FMT, renderer=renderer,
                         frame_size=(geom["out_w"], geom["out_h"]))
        except OSError as e:
            print(f"OpenCV VideoWriter failed to open. {e}")
            return
        n_loop = n_win if n_win > 0 else n_total

        def frames_of(c):
            for _ in range(n_loop):
                ok, fr = c.read()
                if not ok:
                    return
                yield fr

        pos_log = []   # capture position right after frame i was decoded: the batched renderer reads ahead of what it has delivered

        def paired():   # cancel / pause are honoured BEFORE the next pair of frames is decoded (:1196-1222)
            fi, di = frames_of(cap), frames_of(dcap)
            while True:
                while suspend_flag is not None and suspend_flag.is_set() and not cancelled():
                    time.sleep(0.2)
                if cancelled():
                    return
                f, d = next(fi, None), next(di, None)
                if f is None or d is None:
                    return
                pos_log.append(int(cap.get(cv.CAP_PROP_POS_FRAMES)))
                yield f, d

        clip = render_pairs(paired(), renderer=renderer, keep_on_device=sink.nv12, batch=RENDER_BATCH,
                           target_ratio=target_ratio, blank_frames=blank, start_frame_idx=first_idx, skip_first=False,
                           output_height=output_height, fg_shift=fg_shift, mg_shift=mg_shift, bg_shift=bg_shift,
                           sharpness_factor=sharpness_factor, output_format=output_format, dof_strength=dof_strength,
                           feather_strength=feather_strength, blur_ksize=blur_ksize, use_subject_tracking=use_subject_tracking,
                           use_floating_window=use_floating_window, max_pixel_shift_percent=max_pixel_shift_percent,
                           auto_crop_black_bars=auto_crop_black_bars, parallax_balance=parallax_balance,
                           preserve_original_aspect=preserve_original_aspect, zero_parallax_strength=zero_parallax_strength,
                           enable_edge_masking=enable_edge_masking, enable_feathering=enable_feathering,
                           skip_blank_frames=skip_blank_frames, original_video_width=original_video_width,
                           original_video_height=original_video_height, convergence_strength=convergence_strength,
                           enable_dynamic_convergence=enable_dynamic_convergence, ipd_factor=ipd_factor,
                           depth_pop_gamma=depth_pop_gamma, depth_pop_mid=depth_pop_mid, depth_stretch_lo=depth_stretch_lo,
                           depth_stretch_hi=depth_stretch_hi, fg_pop_multiplier=fg_pop_multiplier,
                           bg_push_multiplier=bg_push_multiplier, subject_lock_strength=subject_lock_strength,
                           color_saturation=color_saturation, color_contrast=color_contrast, color_brightness=color_brightness)
        t_prev, rates = time.time(), []
        for i, muxed in enumerate(clip):
            if not sink.write(muxed):
                break
            if end_s is not None and pos_log[i] >= end_idx:   # :1429-1431 with the position the capture had when THIS frame had been read
                break
            now = time.time()
            if now > t_prev:
                rates = (rates + [1.0 / (now - t_prev)])[-10:]
            t_prev = now
            pct = 100.0 * i / max(n_loop, 1)
            if progress is not None:
                progress["value"] = pct
                if hasattr(progress, "update"):
                    progress.update()
            if progress_label is not None:
                avg = sum(rates) / len(rates) if rates else 0.0
                progress_label.config(text=f"{pct:.2f}% | FPS: {avg:.2f}")
        if progress is not None and not cancelled():
            progress["value"] = 100
    except Exception as e:   # the reference reports a crash and still closes its files (:1476-1499)
        print(f"Render crashed: {e}")
    finally:
        if clip is not None and hasattr(clip, "close"):
            clip.close()          # an early break (end_s, sink failure, cancel) must not leave the renderer in overlapped mode until the GC runs
        cap.release()
        dcap.release()
        if sink is not None:
            sink.close(cancelled())
