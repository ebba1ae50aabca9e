"""Procedural INPUTS shared by the golden generator (make_golden.py, development container only) and the tests that replay
the fixtures (CPU oracle tests and the -m gpu tests): no reference import here, so this module travels to the GPU box."""
from __future__ import annotations

import numpy as np

from visiondepth3d_amd import synth


def heal_inputs(case):
    """Procedural inputs of the heal_missing_pixels goldens (shared with tests/test_oracle_vs_golden.py / the GPU test)."""
    H, W, shift, with_edge, hs = case
    f, d = synth.synth_frame(7, H, W)
    orig = (f[..., ::-1].transpose(2, 0, 1).astype(np.float32) / np.float32(255.0)).astype(np.float32)
    warped = np.roll(orig, shift, axis=2).copy()
    warped[:, :, :abs(shift)] = 0.0                      # a disoccluded band, like a forward warp leaves behind
    edge = None
    if with_edge:
        gx = np.zeros_like(d); gx[:, 1:] = np.abs(d[:, 1:] - d[:, :-1])
        edge = np.clip(gx * np.float32(8.0), 0, 1).astype(np.float32)[None]
    return warped, orig, edge, hs


HEAL_CASES = [(96, 160, 3, True, 0.5), (96, 160, 5, False, 0.5), (54, 96, 2, True, 1.0), (33, 47, 1, True, 0.25)]
