import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name))


def golden_json(npz, key):
    return json.loads(bytes(npz[key]).decode())


def u8_diff_stats(a, b):
    d = np.abs(a.astype(np.int16) - b.astype(np.int16))
    return int(d.max()), float(np.count_nonzero(d)) / d.size, float(np.count_nonzero(d > 1)) / d.size


def b2_max_bound(kw):
    """Largest end-to-end difference (in LSB) a <= 1-LSB difference at the warp output (the B1 bar: SLEEF 1-ULP pow / exp in torch vs
    correctly rounded here) can grow to in the muxed frame.  The colour grade scales a channel by up to max(1,sat)*max(1,con) and
    then truncates to uint8 AGAIN -- even the identity grade maps some values v to v-1 (luma mix + two affine steps in float32;
    observed: eyes 63 / 64 -> graded 62 / 64), so one LSB in can be ceil(gain) + 1 out; the sharpen kernel has L1 norm (9+f)/(1+f)
    (core/render_3d.py:719-728), the Dubois anaglyph rows up to 1.43 (:866-883).  The committed fixtures stay within 8."""
    import math
    g = max(1.0, float(kw.get("color_saturation", 1.0))) * max(1.0, float(kw.get("color_contrast", 1.0)))
    f = float(kw.get("sharpness_factor", 0.0))
    b = math.ceil((math.ceil(g - 1e-9) + 1) * (9.0 + f) / (1.0 + f) - 1e-9)
    if kw.get("output_format") == "Red-Cyan Anaglyph":
        b = math.ceil(b * 1.43)
    return b


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    O.lib()
    return O


@pytest.fixture(scope="session")
def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
