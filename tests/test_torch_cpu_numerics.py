"""The oracle's restatement of torch-CPU's transcendental kernels (pow_torch / sigmoid_torch / sqrt_torch / exp_torch in oracle/vd3d_oracle.c)
pinned against torch ITSELF -- the library the reference runs on -- not against a golden file: wherever this torch build dispatches
to the same code paths as the one the fixtures were generated with (AVX-512 ATen kernels = SLEEF 3.6 for pow / sigmoid, oneMKL VML
for sqrt; see the header of each function) every sample must agree bit for bit.  Sizes are multiples of 32: ATen runs a scalar loop
on the last n mod 32 elements of a worker's chunk, which calls libm instead (documented, not reproduced; no video frame size has a
tail).  tools/verify_sleef.py holds the exhaustive offline sweeps (every float32 of [2^-40, 1] for the exponent 0.85 ...)."""
import numpy as np
import pytest


def _torch_avx512_mkl():
    try:
        import torch
        return torch.backends.cpu.get_cpu_capability() == "AVX512" and torch.backends.mkl.is_available()
    except Exception:
        return False


pytestmark = pytest.mark.skipif(not _torch_avx512_mkl(), reason="torch CPU kernels of a different ISA level: other SLEEF / MKL code paths")
N = 1 << 21


@pytest.fixture(autouse=True)
def _one_torch_thread():
    """One worker = one chunk = no scalar tails inside the arrays (their sizes are multiples of 32)."""
    import torch
    n = torch.get_num_threads()
    torch.set_num_threads(1)
    yield
    torch.set_num_threads(n)


def _bits(lo, hi, n, rng):
    return rng.integers(np.float32(lo).view(np.uint32), np.float32(hi).view(np.uint32), n, dtype=np.uint32).view(np.float32).copy()


@pytest.mark.parametrize("gamma", [0.85, 1.5, 0.7, 0.75, 0.9, 0.999, 1.1, 1.2, 1.3, 0.6, 2.2, 0.5, 2.0, 3.0, 1.0, 0.0, -1.0, -2.0])
def test_pow_equals_torch(oracle, gamma):
    """torch.pow(tensor, python float): _signed_pow core/render_3d.py:517 (gamma 0.85 by default, the GUI slider spans 0.7 .. 1.2) and
    the foreground layer weight :620 (1.5).  Uniform samples of [0, 1] plus log-uniform ones down to 2^-40 (SLEEF's double-float
    logarithm loses accuracy below; the shaped depth's |t| is 0 or >= 2^-26)."""
    import torch
    rng = np.random.default_rng(int(abs(gamma) * 1000) + 7 + (gamma < 0))
    x = np.concatenate([rng.uniform(0, 1, N).astype(np.float32), _bits(2.0 ** -40, 1.0, N, rng), np.array([0.0, 1.0] * 16, np.float32)])
    if gamma < 0:
        x = x[x > 0]
        x = x[: x.size // 32 * 32]
    exp = torch.pow(torch.from_numpy(x), gamma).numpy()
    got = oracle.torch_math("pow", x, gamma)
    assert np.array_equal(got.view(np.uint32), exp.view(np.uint32)), int((got != exp).sum())


def test_sigmoid_equals_torch(oracle):
    """torch.sigmoid, suppress_artifacts_with_edge_mask core/render_3d.py:209."""
    import torch
    rng = np.random.default_rng(11)
    x = np.concatenate([rng.uniform(-30, 30, N), rng.uniform(-110, 110, N), rng.normal(0, 1e-3, N)]).astype(np.float32)
    exp = torch.sigmoid(torch.from_numpy(x)).numpy()
    got = oracle.torch_math("sigmoid", x)
    assert np.array_equal(got.view(np.uint32), exp.view(np.uint32)), int((got != exp).sum())


def test_sqrt_equals_torch_and_is_not_the_rounded_root(oracle):
    """torch.sqrt, core/render_3d.py:206 / :349 / :440: MKL's vsSqrt.  One ULP below the correctly rounded root on ~0.6 % of inputs."""
    import torch
    rng = np.random.default_rng(13)
    x = np.concatenate([rng.uniform(0, 2, N).astype(np.float32), _bits(2.0 ** -100, 3.0e38, N, rng),
                        np.array([0.0, 1.0, 4.0, 0.25, 2.0, np.inf, 16.0, 1e-45] * 4, np.float32)])
    exp = torch.sqrt(torch.from_numpy(x)).numpy()
    got = oracle.torch_math("sqrt", x)
    assert np.array_equal(got.view(np.uint32), exp.view(np.uint32)), int((got != exp).sum())
    cr = np.sqrt(x.astype(np.float64)).astype(np.float32)
    low = np.count_nonzero(got != cr) / x.size
    assert 3e-3 < low < 9e-3 and np.all(got.view(np.int32)[got != cr] - cr.view(np.int32)[got != cr] == -1), low


def test_rsqrt14_table_reproduces_the_instruction_where_present(oracle):
    """sqrt_torch rests on a restatement of AVX-512's VRSQRT14PS (a 64-entry table + 10-bit linear interpolation): on every float32
    of [1, 4) -- all mantissas, both exponent parities -- sqrt_torch must equal torch.sqrt (which executes the instruction here)."""
    import torch
    for lo in (0x3f800000, 0x40000000):
        x = np.arange(lo, lo + (1 << 23), dtype=np.uint32).view(np.float32).copy()
        exp = torch.sqrt(torch.from_numpy(x)).numpy()
        assert np.array_equal(oracle.torch_math("sqrt", x), exp)


def test_exp_equals_torch_and_is_not_the_rounded_exponential(oracle):
    """torch.exp of a contiguous float32 tensor (the Gaussian window of the DOF levels, torchvision _get_gaussian_kernel1d via
    core/render_3d.py:798-806) is MKL's vsExp: the oracle's exp_torch AND the product's host-side restatement (vd3d_debug_exp_torch, written
    independently: its tables come from their defining formulas in extended precision) must equal it bit for bit -- on every float32 of two
    whole binades of the window's range, on random samples of [-87, 87] and near zero -- and differ from the rounded exponential on ~1 % of
    the inputs of [-8, 0] by exactly one ULP (tools/verify_vsexp.py: the exhaustive sweep of [2^-40, 87], both signs, 7.8e8 inputs)."""
    import ctypes as C
    import torch
    from visiondepth3d_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(17)
    x = np.concatenate([-np.arange(0x40000000, 0x40000000 + (1 << 23), dtype=np.uint32).view(np.float32),     # every float of (-4, -2]
                        -np.arange(0x3f000000, 0x3f000000 + (1 << 23), dtype=np.uint32).view(np.float32),     # every float of (-1, -0.5]
                        rng.uniform(-8.5, 0, N).astype(np.float32), rng.uniform(-87, 87, N).astype(np.float32),
                        -_bits(2.0 ** -40, 8.5, N, rng), _bits(2.0 ** -40, 8.5, N, rng),
                        np.array([0.0, -0.0, -0.5, -2.0, -8.0, 1.0, -1.0, 87.0] * 4, np.float32)])
    x = np.ascontiguousarray(x[: x.size // 32 * 32])
    exp = torch.exp(torch.from_numpy(x)).numpy()
    got = oracle.torch_math("exp", x)
    assert np.array_equal(got.view(np.uint32), exp.view(np.uint32)), int((got != exp).sum())
    host = np.empty_like(x)
    assert L.vd3d_debug_exp_torch(x.ctypes.data_as(C.c_void_p), host.ctypes.data_as(C.c_void_p), x.size) == 0
    assert np.array_equal(host.view(np.uint32), exp.view(np.uint32)), int((host != exp).sum())
    w = x[(x >= -8) & (x <= 0)]
    cr = np.exp(w.astype(np.float64)).astype(np.float32)
    gw = oracle.torch_math("exp", w)
    off = np.count_nonzero(gw != cr) / w.size
    assert 4e-3 < off < 3e-2 and np.all(np.abs(gw.view(np.int32)[gw != cr] - cr.view(np.int32)[gw != cr]) == 1), off


def test_vsexp_tables_follow_their_definition():
    """The 32-entry tables of exp_torch are not free constants: head = RN(2^(j/32)), tail = RN((2^(j/32) - head) / head).  Re-derived here in
    50-digit arithmetic and compared with what the oracle uses (read back through exp_torch at r = 0: exp(j ln2 / 32) is not exact, so the
    check goes through the literal table in the C source instead)."""
    import os
    import re
    from decimal import Decimal, getcontext
    getcontext().prec = 50
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "vd3d_oracle.c")).read()
    tab = {}
    for name in ("VSEXP_TL", "VSEXP_TH"):
        body = re.search(name + r"\[32\] = \{([^}]*)\}", src).group(1)
        tab[name] = np.array([int(v.strip().rstrip("u"), 16) for v in body.split(",")], np.uint32).view(np.float32)
    for j in range(32):
        v = Decimal(2) ** (Decimal(j) / Decimal(32))
        th = np.float32(float(v))
        assert th == tab["VSEXP_TH"][j], j
        assert np.float32(float((v - Decimal(float(th))) / Decimal(float(th)))) == tab["VSEXP_TL"][j], j
