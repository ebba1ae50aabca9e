"""The device restatement of torch-CPU's pow / sigmoid / sqrt (vd_pow_torch / vd_sigmoid_torch / vd_sqrt_torch in
visiondepth3d_amd/csrc/vd3d_dev.h, reached through vd3d_torch_math) against the oracle's independent C restatement, bit for bit, on
dense and wide input sets -- far beyond the ranges the frame-level parity tests exercise.  The oracle itself is pinned against torch
in tests/test_torch_cpu_numerics.py; together: device == oracle == torch-CPU, the library the reference computes with."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
N = 1 << 22


@pytest.fixture(scope="module")
def R():
    from visiondepth3d_amd.render_3d import Renderer
    r = Renderer(0)
    yield r
    r.close()


def _bits(lo, hi, n, rng):
    return rng.integers(np.float32(lo).view(np.uint32), np.float32(hi).view(np.uint32), n, dtype=np.uint32).view(np.float32).copy()


def _same(R, oracle, op, x, param=0.0):
    got = R.torch_math(op, torch.from_numpy(x).cuda(), param).cpu().numpy()
    exp = oracle.torch_math(op, x, param)
    ne = got.view(np.uint32) != exp.view(np.uint32)
    both_nan = np.isnan(got) & np.isnan(exp)
    bad = np.nonzero(ne & ~both_nan)[0]
    assert bad.size == 0, (op, param, bad.size, [(float(x[i]).hex(), float(got[i]).hex(), float(exp[i]).hex()) for i in bad[:4]])


@pytest.mark.parametrize("gamma", [0.85, 1.5, 0.7, 0.9, 1.2, 0.6, 1.3, 2.2, 0.5, 2.0, 3.0, 1.0, 0.0, -0.5, -1.0, -2.0])
def test_pow_device_equals_oracle(R, oracle, gamma):
    rng = np.random.default_rng(int(abs(gamma) * 1000) + (gamma < 0))
    x = np.concatenate([rng.uniform(0, 1, N).astype(np.float32), _bits(2.0 ** -126, 4.0, N, rng), np.array([0.0, 1.0, 0.5, 2.0 ** -24, 1 - 2.0 ** -24], np.float32)])
    if gamma < 0:
        x = x[x > 0]
    _same(R, oracle, "pow", x, gamma)


def test_pow_device_equals_oracle_on_every_float_of_a_binade(R, oracle):
    """gamma 0.85 and 1.5 (the two exponents render_sbs_3d passes) on all 2^23 floats of [0.25, 0.5) and of [2^-12, 2^-11)."""
    for lo in (0x3e800000, 0x39800000):
        x = np.arange(lo, lo + (1 << 23), dtype=np.uint32).view(np.float32).copy()
        _same(R, oracle, "pow", x, 0.85)
        _same(R, oracle, "pow", x, 1.5)


def test_sigmoid_device_equals_oracle(R, oracle):
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.uniform(-30, 30, N), rng.uniform(-120, 120, N), rng.normal(0, 1e-3, N), [0.0, -0.0, 88.0, -88.0, 104.5, -104.5, 1e-40]]).astype(np.float32)
    _same(R, oracle, "sigmoid", x)


def test_sqrt_device_equals_oracle(R, oracle):
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.uniform(0, 2, N).astype(np.float32), _bits(1e-45, 3.0e38, N, rng), np.array([0.0, 1.0, 4.0, 0.25, 2.0, np.inf, 2.0 ** -100, 2.0 ** -101, 1e-45], np.float32)])
    _same(R, oracle, "sqrt", x)
    for lo in (0x3f800000, 0x40000000):        # every mantissa, both exponent parities: the whole VRSQRT14 table
        _same(R, oracle, "sqrt", np.arange(lo, lo + (1 << 23), dtype=np.uint32).view(np.float32).copy())


def test_torch_math_rejects_bad_arguments(R):
    with pytest.raises(KeyError):
        R.torch_math("exp", torch.zeros(4))
    import ctypes as C
    assert R._L.vd3d_torch_math(R._ctx, 7, None, C.c_float(0), None, 4) < 0
