"""Which frames of a clip are blank: the side channel behind ``skip_blank_frames``.

Mirrors the interface of the reference's ``core/ffmpeg_blackdetect.py:23-81`` (same function name, arguments,
cache file and failure behaviour) so ``render_sbs_3d(skip_blank_frames=True)`` keeps working: ffmpeg's
``blackdetect`` filter reports ``black_start:<seconds>`` on stderr, each start time becomes the frame index
``int(seconds * fps)``, and ONLY these start frames are treated as blank (the reference never expands an interval
to its duration -- reproduced as is).  The list is cached next to the video as ``<video>.blankcache.json``.

This is host-side I/O plumbing: the per-frame work for a blank frame is ``vd3d_render_frame_blank``.
"""
from __future__ import annotations

import json
import os
import re
import subprocess

_START = re.compile(r"black_start:(\d+\.\d+)")   # integral seconds print as "12" and are missed, like in the reference (:65)


def parse_blackdetect_log(stderr_text: str, fps: float) -> list[int]:
    """ffmpeg blackdetect stderr -> sorted frame indices (:65-68,77)."""
    return sorted(int(float(t) * fps) for t in _START.findall(stderr_text))


def get_video_fps(input_path: str) -> float:
    """r_frame_rate of the first video stream via ffprobe; 30 when ffprobe is unavailable (:8-21)."""
    cmd = ["ffprobe", "-v", "error", "-select_streams", "v:0", "-show_entries", "stream=r_frame_rate",
           "-of", "default=noprint_wrappers=1:nokey=1", input_path]
    try:
        txt = subprocess.run(cmd, capture_output=True, text=True).stdout.strip()
        num, den = (int(v) for v in txt.split("/"))
        return num / den
    except Exception as e:
        print(f"[Warning] Failed to get FPS with ffprobe: {e}")
        return 30


def blackdetect_filter(mode: str, duration_threshold: float, pixel_threshold: float) -> str:
    if mode == "black":
        return f"blackdetect=d={duration_threshold}:pix_th={pixel_threshold}"
    if mode == "white":
        # the reference's white mode is a raw (non-f) string, so the thresholds are NOT substituted (:51): the literal text is kept
        return r"lutrgb='r=max(val\,240):g=max(val\,240):b=max(val\,240)',blackdetect=d={duration_threshold}:pix_th={pixel_threshold}"
    raise ValueError("mode must be 'black' or 'white'")


def detect_black_white_frames(input_path, mode="black", duration_threshold=0.1, pixel_threshold=0.10, cache=True):
    cache_file = input_path + ".blankcache.json"
    if cache and os.path.exists(cache_file):
        try:
            with open(cache_file) as f:
                return json.load(f)
        except Exception:
            pass
    fps = get_video_fps(input_path)
    vf = blackdetect_filter(mode, duration_threshold, pixel_threshold)
    try:
        log = subprocess.run(["ffmpeg", "-i", input_path, "-vf", vf, "-an", "-f", "null", "-"],
                             capture_output=True, text=True).stderr
    except Exception as e:
        print(f"[Warning] FFmpeg frame detect failed: {e}")
        return []
    frames = [int(float(t) * fps) for t in _START.findall(log)]
    if cache:
        try:
            with open(cache_file, "w") as f:
                json.dump(frames, f)   # unsorted on disk, sorted on return (:70-77)
        except Exception:
            pass
    return sorted(frames)
