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


def _same_aten(R, oracle, op, x, param, threads, allow=0):
    got = R.torch_math_aten(op, torch.from_numpy(x).cuda(), param, threads).cpu().numpy()
    exp = oracle.torch_math_aten({"pow": 0, "sigmoid": 1}[op], x, param, threads)
    bad = np.nonzero((got.view(np.uint32) != exp.view(np.uint32)) & ~(np.isnan(got) & np.isnan(exp)))[0]
    assert bad.size <= allow, (op, param, threads, bad.size, [(float(x[i]).hex(), float(got[i]).hex(), float(exp[i]).hex()) for i in bad[:4]])
    return bad.size


def test_glibc_expf_tail_of_sigmoid_device_equals_oracle(R, oracle):
    """Round 5, ATen's scalar tails: torch.sigmoid on the last (chunk length mod 32) elements of a thread's chunk is 1 / (1 + expf(-x)) with GLIBC's expf,
    restated in double arithmetic on the device (vd_expf_glibc) and in the oracle (checked there against libm on all 2^32 inputs).  Every element through
    the tail arithmetic (aten_threads < 0): exact, incl. the overflow / underflow branches and subnormal results."""
    rng = np.random.default_rng(13)
    x = np.concatenate([rng.uniform(-30, 30, N), rng.uniform(-120, 120, N), rng.normal(0, 1e-3, N), _bits(1e-30, 100.0, N, rng), -_bits(1e-30, 110.0, N, rng),
                        [0.0, -0.0, 88.0, -88.0, 88.72, 88.73, -103.9, -104.0, 104.5, -104.5, 1e-40, np.inf, -np.inf, 32.5647, -63.0994]]).astype(np.float32)
    _same_aten(R, oracle, "sigmoid", x, 0.0, -1)
    for lo in (0x42000000, 0xc2700000):     # every float of the binades holding the two inputs on which glibc's SSE2 and FMA builds differ (0x4202422f, 0xc27c65d9)
        _same_aten(R, oracle, "sigmoid", np.arange(lo, lo + (1 << 23), dtype=np.uint32).view(np.float32).copy(), 0.0, -1)


@pytest.mark.parametrize("gamma", [0.85, 1.5, 0.7, 1.17, 0.6123, 1.3])
def test_libm_pow_tail_device_equals_oracle(R, oracle, gamma):
    """torch.pow on a scalar tail is (float) std::pow((double) x, gamma): the oracle calls glibc's pow, the device ocml's.  Both are within an ULP of a DOUBLE, so
    the float32 results agree unless x^gamma lies within ~2^-52 of a float32 rounding boundary -- expected on about one element in 2^28; none of these 8.4 M."""
    rng = np.random.default_rng(int(gamma * 1000))
    x = np.concatenate([rng.uniform(0, 1, N).astype(np.float32), _bits(2.0 ** -126, 4.0, N, rng), np.array([0.0, 1.0, 0.5, 2.0 ** -24, 1 - 2.0 ** -24, 1e-45], np.float32)])
    _same_aten(R, oracle, "pow", x, gamma, -1, allow=1)


@pytest.mark.parametrize("n,threads", [(31, 4), (3500, 8), (32768, 8), (32769, 8), (100003, 3), (123 * 457, 5), (333 * 777, 16), (1000003, 64), (1920 * 1080, 8), (250 * 333, 1)])
def test_tail_positions_device_equals_oracle(R, oracle, n, threads):
    """WHERE the tails are: min(threads, ceil(n / 32768)) chunks of ceil(n / that), the last (length mod 32) elements of each (vd_tails_of == the oracle's
    at_tails_of, which tests/test_aten_restatements.py pins against torch).  1920 x 1080 at 8 threads has none."""
    rng = np.random.default_rng(n)
    x = rng.uniform(0, 1, n).astype(np.float32)
    _same_aten(R, oracle, "pow", x, 0.85, threads)
    _same_aten(R, oracle, "sigmoid", (x * 24 - 12).astype(np.float32), 0.0, threads)
    got = R.torch_math_aten("pow", torch.from_numpy(x).cuda(), 0.85, threads).cpu().numpy()
    plain = R.torch_math("pow", torch.from_numpy(x).cuda(), 0.85).cpu().numpy()
    if n == 1920 * 1080:
        assert np.array_equal(got, plain)
    elif n == 3500:
        assert 0 < np.count_nonzero(got != plain) <= 12 and np.array_equal(got[:3488], plain[:3488])      # one chunk: the last 3500 mod 32 = 12 elements (~28 % of them differ)


def test_torch_math_rejects_bad_arguments(R):
    with pytest.raises(KeyError):
        R.torch_math("exp", torch.zeros(4))
    import ctypes as C
    assert R._L.vd3d_torch_math(R._ctx, 7, None, C.c_float(0), None, 4) < 0
