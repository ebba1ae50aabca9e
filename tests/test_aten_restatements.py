"""The oracle's restatements of the ATen CPU kernels the reference's float32 values depend on, pinned against torch ITSELF (no reference
tree needed): F.avg_pool2d's window-sum order, the bilinear F.interpolate, F.grid_sample(bilinear, border, align_corners=True),
torch.linspace and torch.quantile.  Bit for bit -- these are the choices (FMA contraction, summation order, rank arithmetic) that
decide the last bit of the reference's planes; tests/test_torch_cpu_numerics.py does the same for pow / sigmoid / sqrt."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
F = torch.nn.functional


@pytest.fixture(autouse=True)
def _threads():
    n = torch.get_num_threads()
    torch.set_num_threads(4)
    yield
    torch.set_num_threads(n)


@pytest.mark.parametrize("k", [3, 5, 9, 15])
@pytest.mark.parametrize("shape", [(50, 70), (108, 192), (270, 480)])
def test_avg_pool2d_is_one_running_sum_row_major(oracle, k, shape):
    """core/render_3d.py:213, 355, 444, 456.  ATen's cpu_avg_pool2d adds the window row-major into ONE float32 accumulator and divides by
    k * k (count_include_pad): a separable or pairwise sum differs in the last bit of most outputs."""
    x = np.random.default_rng(k * 1000 + shape[0]).random(shape, dtype=np.float32)
    exp = F.avg_pool2d(torch.from_numpy(x)[None, None], k, stride=1, padding=k // 2)[0, 0].numpy()
    assert np.array_equal(oracle.avg_pool2d(x, k), exp)


@pytest.mark.parametrize("src,dst", [((54, 96), (108, 192)), ((108, 192), (108, 192)), ((270, 480), (540, 960)), ((100, 77), (131, 203))])
def test_bilinear_interpolate(oracle, src, dst):
    """core/render_3d.py:595-596 (align_corners=False), planes of >= 4 K elements (below that ATen switches to a variant with premultiplied
    weights, one ULP away: documented in test_oracle_vs_live_reference.py)."""
    x = np.random.default_rng(src[0] + dst[1]).random((3,) + src, dtype=np.float32)
    exp = F.interpolate(torch.from_numpy(x)[None], size=dst, mode="bilinear", align_corners=False)[0].numpy()
    assert np.array_equal(oracle.interp_bilinear(x, dst[0], dst[1]), exp)


@pytest.mark.parametrize("shape", [(72, 128), (108, 192), (270, 480)])
def test_grid_sample_bilinear_border_align_corners(oracle, shape):
    """core/render_3d.py:697-701: the warp.  Grids built like the reference's (linspace mesh + a horizontal shift)."""
    H, W = shape
    rng = np.random.default_rng(H)
    plane = rng.random(shape, dtype=np.float32)
    ys = torch.linspace(-1, 1, H)
    xs = torch.linspace(-1, 1, W)
    gy, gx = torch.meshgrid(ys, xs, indexing="ij")
    shift = torch.from_numpy(((rng.random(shape, dtype=np.float32) - 0.5) * 0.08).astype(np.float32))
    grid = torch.stack((gx + shift, gy), dim=-1)
    exp = F.grid_sample(torch.from_numpy(plane)[None, None], grid[None], mode="bilinear", padding_mode="border", align_corners=True)[0, 0].numpy()
    assert np.array_equal(oracle.grid_sample(plane, grid.numpy()), exp)


@pytest.mark.parametrize("steps", [2, 3, 108, 192, 1080, 1920, 3840])
def test_linspace(oracle, steps):
    assert np.array_equal(oracle.linspace(-1.0, 1.0, steps), torch.linspace(-1, 1, steps).numpy())
    assert np.array_equal(oracle.linspace(0.0, 2.0, steps), torch.linspace(0.0, 2.0, steps).numpy())


@pytest.mark.parametrize("q", [0.02, 0.05, 0.5, 0.95, 0.98])
def test_quantile(oracle, q):
    """core/render_3d.py:249-250, 536-537: torch.quantile's linear interpolation between the two order statistics, in float32."""
    rng = np.random.default_rng(int(q * 100))
    for n in (1000, 20736, 129600):
        v = rng.random(n, dtype=np.float32)
        if n == 20736:
            v = (np.floor(v * 255) / 255).astype(np.float32)       # an 8-bit depth plane: many ties
        assert np.float32(oracle.quantile(v, q)) == np.float32(torch.quantile(torch.from_numpy(v), q).item()), (q, n)
