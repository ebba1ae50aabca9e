"""Import the reference's ``core/render_3d.py`` by path (development container only).

``/root/reference`` never travels to the GPU box, so nothing under ``-m gpu``,
``smoke()`` or ``bench.py`` may import this module.  It exists to (a) generate
the committed golden vectors (``make_golden.py``) and (b) let the CPU-only
oracle tests cross-check the oracle against the live reference when the
reference tree happens to be present.
"""
from __future__ import annotations

import importlib.util
import io
import contextlib
import os
import sys
import types

REF_ROOT = os.environ.get("VD3D_REFERENCE", "/root/reference")

_cached = None


def available() -> bool:
    return os.path.isfile(os.path.join(REF_ROOT, "core", "render_3d.py"))


def load():
    """Return the reference ``core.render_3d`` module (stubs installed, ``core/__init__`` bypassed)."""
    global _cached
    if _cached is not None:
        return _cached
    if not available():
        raise RuntimeError("reference tree not present")
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import ref_stubs  # noqa: E402

    ref_stubs.install()
    core = types.ModuleType("core")
    core.__path__ = [os.path.join(REF_ROOT, "core")]
    sys.modules["core"] = core

    def _load(name):
        spec = importlib.util.spec_from_file_location(f"core.{name}", os.path.join(REF_ROOT, "core", f"{name}.py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[f"core.{name}"] = mod
        with contextlib.redirect_stdout(io.StringIO()):
            spec.loader.exec_module(mod)
        return mod

    _load("ffmpeg_blackdetect")
    _cached = _load("render_3d")
    return _cached


def load_preview_utils():
    """The reference's ``core/preview_utils.py`` (needs ``core.render_3d`` loaded first)."""
    load()
    name = "preview_utils"
    if f"core.{name}" in sys.modules:
        return sys.modules[f"core.{name}"]
    spec = importlib.util.spec_from_file_location(f"core.{name}", os.path.join(REF_ROOT, "core", f"{name}.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[f"core.{name}"] = mod
    with contextlib.redirect_stdout(io.StringIO()):
        spec.loader.exec_module(mod)
    return mod


def reset_state(r3d=None):
    """Reset the module-level tracker singletons."""
    r3d = r3d or load()
    r3d.floating_window_tracker.prev_offset = 0.0
    r3d.floating_window_tracker.frame_counter = 0
    r3d.depth_ema_norm._lo = None
    r3d.depth_ema_norm._hi = None
    r3d.conv_ema.val = None
    r3d.bar_easer.prev_bar_width = 0
    r3d.global_session_start_time = None
