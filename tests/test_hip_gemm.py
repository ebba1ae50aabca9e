"""GPU tests (-m gpu) of the split-bf16 GEMM (vd3d_gemm_x3, csrc/vd3d_gemm.hip): the opt-in `gemm="bf16x3"` mode of the depth leg (round 6).

A floating-point kernel: the bar is stated against FLOAT64, beside torch's own float32 GEMM (hipBLASLt) on the same operands --
  max over the elements of |y - y64| / (sum_k |x||w| + |b|) <= 1.25 x the float32 library GEMM's own figure (which grows with K like any float32 sum),
  relative RMS error <= 1.25 x the float32 library GEMM's.
Measured on MI355X at M = 39 088 (tools/probe_gemm_x3.py, round 6): 2.6 - 3.5e-7 against 3.5 - 4.6e-7 for the float32 GEMM, RMS 0.86x of it -- the six exact
products accumulated in float32 are, if anything, slightly MORE accurate than hipBLASLt's float32 kernel (whose MFMA sums in chunks of its own).
Then the depth leg in that mode against the STOCK float32 Hugging Face graph with the bar the float32 leg itself meets (tests/test_hip_depth_e2e.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
F = torch.nn.functional


@pytest.fixture(scope="module")
def R():
    from visiondepth3d_amd.render_3d import Renderer
    assert torch.cuda.is_available()
    r = Renderer(0)
    yield r
    r.close()


def _err(y, x, w, b, gelu):
    ref = F.linear(x.double(), w.double(), None if b is None else b.double())
    scale = x.abs().double() @ w.abs().double().T
    if b is not None:
        scale = scale + b.abs().double()
    if gelu:
        ref = F.gelu(ref)   # |gelu'| <= 1.13: the same bar holds behind it (+ erff's own 1-2 ULP, inside the factor 4)
        scale = scale * 1.13 + 1e-30
    d = (y.double() - ref).abs()
    return float((d / scale).max()), float(d.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())


@pytest.mark.parametrize("mode", ["bf16x3", "fp16x2"])
@pytest.mark.parametrize("M,K,N,bias,gelu", [(1000, 768, 2304, True, False), (257, 768, 768, True, False), (2443, 768, 3072, True, True),
                                             (600, 3072, 768, True, False), (513, 384, 1152, True, False), (300, 1024, 4096, False, True),
                                             (1, 16, 1, True, False), (255, 48, 100, False, False), (256, 32, 256, True, False)])
def test_gemm_x3_is_float32_faithful(R, M, K, N, bias, gelu, mode):
    """bf16x3: the bars of the module docstring.  fp16x2 (22-bit operands, half the matrix work): RMS <= 1.5 x and maximum <= 2 x the float32 library GEMM's for
    K >= 64 -- its extra per-product error (<= 3 x 2^-22) sits under the float32 accumulation error both share once a few dozen products are summed; for the two
    tiny-K cases only the absolute bar |y - y64| <= 2^-20 (sum |x||w| + |b|) applies (nothing to hide behind)."""
    g = torch.Generator(device="cuda").manual_seed(M * 7 + K)
    # operands with a wide dynamic range (LayerNorm outputs times a few outlier channels, like DINOv2's residual stream)
    x = torch.randn(M, K, device="cuda", generator=g) * torch.exp(torch.randn(1, K, device="cuda", generator=g) * 1.5)
    w = torch.randn(N, K, device="cuda", generator=g) * 0.05
    b = torch.randn(N, device="cuda", generator=g) if bias else None
    img = R.gemm_x3_pack(w, mode)
    y = R.linear_x3(x, img, N, b, gelu=gelu, mode=mode)
    assert y.shape == (M, N) and y.dtype == torch.float32 and bool(torch.isfinite(y).all())
    y32 = F.linear(x, w, b)
    if gelu:
        y32 = F.gelu(y32)
    e3, r3 = _err(y, x, w, b, gelu)
    e32, r32 = _err(y32, x, w, b, gelu)
    if mode == "bf16x3":
        assert e3 <= max(1.25 * e32, 2.0 ** -24), (e3, e32)
        assert r3 <= 1.25 * r32 + 1e-9, (r3, r32)
    elif K >= 64:
        assert e3 <= max(2.0 * e32, 2.0 ** -22), (e3, e32)
        assert r3 <= 1.5 * r32 + 1e-9, (r3, r32)
    else:
        assert e3 <= 2.0 ** -20, (e3, e32)
    # a batch dimension in front, and the same call again (no state between calls)
    if M % 2 == 0:
        y2 = R.linear_x3(x.view(2, M // 2, K), img, N, b, gelu=gelu, mode=mode)
        assert y2.shape == (2, M // 2, N) and torch.equal(y2.view(M, N), y)
    assert torch.equal(R.linear_x3(x, img, N, b, gelu=gelu, mode=mode), y)


@pytest.mark.parametrize("mode", ["bf16x3", "fp16x2"])
@pytest.mark.parametrize("M,K,N,gelu", [(21920, 768, 768, False), (21920, 768, 2304, False), (21920, 768, 3072, True), (21920, 3072, 768, False), (21000, 256, 1000, False)])
def test_gemm_x3_at_the_depth_legs_batch_size(R, M, K, N, gelu, mode):
    """The four linears at the size the benchmark runs them (16 frames x 1 370 tokens of DA-V2-Base at 4K: 86 M tiles, 11 per XCD -- every XCD's last round of tiles is
    a short one) and one ragged shape: the bars of test_gemm_x3_is_float32_faithful per block of 4 096 rows (so that a failure names the tile rows), and identical bits
    from two calls.  (Round 6 tried handing an XCD's short last round to split-K slice workgroups -- tools/r06/gemm_tail_splitk.patch -- and measured it SLOWER:
    a tile that runs with most of the chip idle takes a third of the time of one that shares it, so the short round was never the cost the tile count suggests.)"""
    g = torch.Generator(device="cuda").manual_seed(M + K)
    x = torch.randn(M, K, device="cuda", generator=g) * torch.exp(torch.randn(1, K, device="cuda", generator=g) * 1.5)
    w = torch.randn(N, K, device="cuda", generator=g) * 0.05
    b = torch.randn(N, device="cuda", generator=g)
    img = R.gemm_x3_pack(w, mode)
    y = R.linear_x3(x, img, N, b, gelu=gelu, mode=mode)
    y32 = F.linear(x, w, b)
    if gelu:
        y32 = F.gelu(y32)
    for m0 in range(0, M, 4096):
        sl = slice(m0, min(m0 + 4096, M))
        e3, r3 = _err(y[sl], x[sl], w, b, gelu)
        e32, r32 = _err(y32[sl], x[sl], w, b, gelu)
        f_max, f_rms = (1.25, 1.25) if mode == "bf16x3" else (2.0, 1.5)
        assert e3 <= max(f_max * e32, 2.0 ** -22), (m0, e3, e32)
        assert r3 <= f_rms * r32 + 1e-9, (m0, r3, r32)
    assert torch.equal(R.linear_x3(x, img, N, b, gelu=gelu, mode=mode), y)


def test_gemm_x3_split_is_exact_and_layout_is_asymmetric(R):
    """Known answers: (i) with W = I the GEMM returns x itself BIT FOR BIT (x = x1 + x2 + x3 exactly, each term times 1.0, small terms first) -- for values across
    the exponent range, negative numbers and zeros; (ii) an asymmetric integer-valued problem, exact in every arithmetic: catches a transposed / permuted tile."""
    K = N = 256
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(300, K, device="cuda", generator=g) * torch.exp(torch.randn(300, K, device="cuda", generator=g) * 8.0)
    x[0, :8] = torch.tensor([0.0, -0.0, 1.0, -1.0, 3.0e-30, 1.0e30, 1.1754944e-38, 16777215.0], device="cuda")
    img = R.gemm_x3_pack(torch.eye(K, device="cuda"))
    y = R.linear_x3(x, img, N)
    assert torch.equal(y, x)
    M, K, N = 515, 64, 320
    xi = (torch.arange(M * K, device="cuda").view(M, K) % 13 - 6).float()
    wi = ((torch.arange(N * K, device="cuda").view(N, K) * 7) % 11 - 5).float()
    bi = torch.arange(N, device="cuda").float()
    y = R.linear_x3(xi, R.gemm_x3_pack(wi), N, bi)
    assert torch.equal(y.double(), F.linear(xi.double(), wi.double(), bi.double()))
    # small integers are exact in fp16 too (second terms zero, the row scale a power of two): the fp16x2 form must give the same exact answer
    y = R.linear_x3(xi, R.gemm_x3_pack(wi, "fp16x2"), N, bi, mode="fp16x2")
    assert torch.equal(y.double(), F.linear(xi.double(), wi.double(), bi.double()))


@pytest.mark.parametrize("mode", ["bf16x3", "fp16x2"])
@pytest.mark.parametrize("B,T,H", [(1, 64, 1), (2, 77, 3), (1, 300, 2), (2, 1370, 6), (1, 2443, 12), (3, 257, 12), (1, 31, 2)])
def test_attention_x3_is_float32_faithful(R, B, T, H, mode):
    """vd3d_attention_x3 against float64 softmax attention, beside PyTorch's float32 scaled_dot_product_attention (AOTriton) on the same operands: RMS error <= 1.25 x the
    float32 kernel's (measured: 0.87 x), the maximum over all elements -- one sample of a noisy tail -- <= 2 x the float32 kernel's (measured: 0.9 - 1.6 x).  Token counts that are not multiples of the 64-row KV tile / 32-query block / 256-query
    workgroup, DINOv2's 1370 (1080p) and 2443 (4K) among them; logits with a wide range (a softmax that is nearly one-hot for some queries, flat for others)."""
    D = 64
    g = torch.Generator(device="cuda").manual_seed(B * 1000 + T)
    qkv = torch.randn(B, T, 3, H, D, device="cuda", generator=g)
    qkv[:, :, 0] *= torch.exp(torch.randn(B, T, H, 1, device="cuda", generator=g))        # per-query temperature
    qkv[:, :, 2] *= 3.0
    scale = D ** -0.5
    out = R.attention_x3(qkv.view(B, T, 3 * H * D), H, scale, mode=mode)
    assert out.shape == (B, T, H * D) and bool(torch.isfinite(out).all())
    q, k, v = (qkv[:, :, i].transpose(1, 2) for i in range(3))
    ref = torch.softmax((q.double() @ k.double().transpose(-1, -2)) * scale, dim=-1) @ v.double()
    ref = ref.transpose(1, 2).reshape(B, T, H * D)
    o32 = F.scaled_dot_product_attention(q, k, v, scale=scale).transpose(1, 2).reshape(B, T, H * D)
    e3, e32 = float((out.double() - ref).abs().max()), float((o32.double() - ref).abs().max())
    r3 = float((out.double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
    r32 = float((o32.double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
    kmax, krms = (2.0, 1.25) if mode == "bf16x3" else (3.0, 2.0)   # fp16x2: the logits are 64-term sums of 22-bit products: their error shows through the exponential
    assert e3 <= max(kmax * e32, 1e-6 * float(ref.abs().max())), (e3, e32)
    assert r3 <= krms * r32 + 1e-8, (r3, r32)
    assert torch.equal(R.attention_x3(qkv.view(B, T, 3 * H * D), H, scale, mode=mode), out)   # no state between calls


def test_attention_x3_full_batch_is_deterministic_under_load(R):
    """A batch that fills the chip several times over (16 frames x 12 heads x 1370 tokens = 1 152 workgroups), five times in a row: identical bits every time, and
    frame 0 equal to the same frame computed alone.  (The first software-pipelined version of the kernel let the DMA of K tile 3 overwrite K tile 0 while slower
    waves were still reading it -- no barrier behind the prologue's S^T -- which showed on exactly this shape, under load, and nowhere in the small cases.)"""
    B, T, H, D = 16, 1370, 12, 64
    g = torch.Generator(device="cuda").manual_seed(11)
    qkv = torch.randn(B, T, 3 * H * D, device="cuda", generator=g)
    for mode in ("bf16x3", "fp16x2"):
        out = R.attention_x3(qkv, H, 0.125, mode=mode)
        for _ in range(5):
            assert torch.equal(R.attention_x3(qkv, H, 0.125, mode=mode), out)
        assert torch.equal(R.attention_x3(qkv[:1].contiguous(), H, 0.125, mode=mode), out[:1])


def test_attention_x3_known_answers(R):
    """(i) one key: softmax is 1, the output is v itself -- bit for bit (v = v1 + v2 + v3 exactly, p = 1).  (ii) identical keys: the output is the mean of the values
    (uniform probabilities 1 / T with T a power of two: exact).  (iii) a query that matches one key by a wide margin copies that key's value."""
    D, H = 64, 2
    g = torch.Generator(device="cuda").manual_seed(3)
    qkv = torch.randn(2, 1, 3, H, D, device="cuda", generator=g)
    out = R.attention_x3(qkv.view(2, 1, -1), H, 0.125)
    assert torch.equal(out.view(2, 1, H, D), qkv[:, :, 2])
    T = 128
    qkv = torch.randn(1, T, 3, H, D, device="cuda", generator=g)
    qkv[:, :, 1] = qkv[:, :1, 1]                                   # every key the same -> uniform softmax
    vi = torch.randint(-8, 9, (1, T, H, D), device="cuda", generator=g).float()   # integer values: their mean over 128 tokens is exact in float32
    qkv[:, :, 2] = vi
    out = R.attention_x3(qkv.view(1, T, -1), H, 0.125).view(1, T, H, D)
    assert torch.equal(out, vi.mean(dim=1, keepdim=True).expand(1, T, H, D))
    T = 200
    qkv = torch.randn(1, T, 3, H, D, device="cuda", generator=g) * 0.01
    e = torch.zeros(D, device="cuda"); e[5] = 1.0
    qkv[0, :, 0] = e * 64.0                                        # every query points at key 17 with logit 64 * 64 * 0.125 = 512 above the rest
    qkv[0, 17, 1] = e * 64.0
    out = R.attention_x3(qkv.view(1, T, -1), H, 0.125).view(1, T, H, D)
    assert torch.allclose(out, qkv[:, 17:18, 2].expand(1, T, H, D), rtol=0, atol=1e-30)


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(1, 16, 32, 16, 64), (2, 37, 66, 128, 128), (1, 50, 45, 96, 128), (3, 19, 33, 64, 64), (1, 74, 132, 128, 64),
                                              (1, 1, 1, 32, 64), (2, 17, 31, 48, 128), (2, 45, 70, 64, 32), (1, 16, 32, 16, 32)])
def test_conv3x3_x2_is_float32_faithful(R, B, H, W, Cin, Cout):
    """vd3d_conv3x3_x2 (the DPT neck / head convolutions in the fp16x2 arithmetic) against a float64 convolution, beside the float32 library convolution on the same
    operands: RMS error <= 1.5 x, maximum <= 2.5 x the library's.  Sizes that are not multiples of the 16 x 32 tile, a one-pixel image (every tap but the centre is
    padding), the neck's 96-channel input; post-ReLU inputs with a few large channels, like the maps the fusion stage sees."""
    g = torch.Generator(device="cuda").manual_seed(B * 100 + H)
    x = torch.relu(torch.randn(B, Cin, H, W, device="cuda", generator=g)) * torch.exp(torch.randn(1, Cin, 1, 1, device="cuda", generator=g))
    x = x.contiguous(memory_format=torch.channels_last)
    w = torch.randn(Cout, Cin, 3, 3, device="cuda", generator=g) * 0.05
    img = R.conv3x3_x2_pack(w)
    assert img is not None
    y = R.conv3x3_x2(x, img, Cout)
    assert y.shape == (B, Cout, H, W) and y.is_contiguous(memory_format=torch.channels_last) and bool(torch.isfinite(y).all())
    ref = F.conv2d(x.double(), w.double(), None, 1, 1)
    y32 = F.conv2d(x, w, None, 1, 1)
    scale = F.conv2d(x.abs().double(), w.abs().double(), None, 1, 1) + 1e-30
    e3, e32 = float(((y.double() - ref).abs() / scale).max()), float(((y32.double() - ref).abs() / scale).max())
    r3 = float((y.double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
    r32 = float((y32.double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
    assert e3 <= max(2.5 * e32, 2.0 ** -21), (e3, e32)
    assert r3 <= 1.5 * r32 + 1e-9, (r3, r32)
    assert torch.equal(R.conv3x3_x2(x, img, Cout), y)
    # exact on small integers (fp16-exact operands, power-of-two channel scales): catches a transposed tap, a shifted halo or a swapped channel half
    xi = ((torch.arange(B * Cin * H * W, device="cuda").view(B, Cin, H, W) * 7) % 5 - 2).float().contiguous(memory_format=torch.channels_last)
    wi = ((torch.arange(Cout * Cin * 9, device="cuda").view(Cout, Cin, 3, 3) * 11) % 7 - 3).float()
    yi = R.conv3x3_x2(xi, R.conv3x3_x2_pack(wi), Cout)
    assert torch.equal(yi.double(), F.conv2d(xi.double(), wi.double(), None, 1, 1))


def test_conv3x3_x2_refuses_shapes_it_does_not_build(R):
    assert R.conv3x3_x2_pack(torch.zeros(16, 64, 3, 3, device="cuda")) is None      # C_out 16 (32 / 64 / 128 are built)
    assert R.conv3x3_x2_pack(torch.zeros(64, 20, 3, 3, device="cuda")) is None      # C_in not a multiple of 16
    assert R.conv3x3_x2_pack(torch.zeros(64, 64, 1, 1, device="cuda")) is None      # not 3 x 3


def test_gemm_x3_argument_checks(R):
    from visiondepth3d_amd._lib import Vd3dError
    with pytest.raises(NotImplementedError):
        R.gemm_x3_pack(torch.zeros(8, 20, device="cuda"))        # K not a multiple of 16
    img = R.gemm_x3_pack(torch.zeros(8, 32, device="cuda"))
    with pytest.raises(Vd3dError):
        R.linear_x3(torch.zeros(4, 24, device="cuda"), img, 8)    # K mismatch shows as an unsupported K


@pytest.mark.parametrize("mode", ["bf16x3", "fp16x2"])
def test_depth_leg_bf16x3_meets_the_float32_legs_bar_1080p(R, mode):
    """The acceptance the float32 leg itself meets against the STOCK float32 graph (tests/test_hip_depth_e2e.py): raw prediction within 1e-4 of its range,
    >= 99.5 % of the uint8 plane's bytes identical, no byte off by more than one level -- with every transformer linear on the split-bf16 GEMM."""
    from test_hip_depth_e2e import _plane_stats, _record, _stock_u8_planes
    from visiondepth3d_amd import synth
    from visiondepth3d_amd.depth import DepthPipe
    H, W = 1080, 1920
    frames = torch.from_numpy(np.stack([synth.synth_frame(i, H, W)[0] for i in range(2)])).cuda()
    exp_u8, exp_pred = _stock_u8_planes("depth-anything-v2-small", frames)
    pipe = DepthPipe("depth-anything-v2-small", device="cuda", dtype=torch.float32, renderer=R, gemm=mode)
    pred = pipe.infer_bgr_u8(frames, raw=True)
    st = _plane_stats(R.depth_handoff(pred, H, W), exp_u8)
    st["pred_max_err_of_range"] = float((pred - exp_pred).abs().max()) / float(exp_pred.max() - exp_pred.min())
    # the float32 leg on the same frames, for the record: the two modes side by side
    p32 = DepthPipe("depth-anything-v2-small", device="cuda", dtype=torch.float32, renderer=R).infer_bgr_u8(frames, raw=True)
    st["f32_leg_pred_max_err_of_range"] = float((p32 - exp_pred).abs().max()) / float(exp_pred.max() - exp_pred.min())
    st["x3_vs_f32_leg_u8"] = _plane_stats(R.depth_handoff(pred, H, W), R.depth_handoff(p32, H, W))
    _record(mode + "_1080p", st)
    assert st["pred_max_err_of_range"] < 1e-4, st
    assert st["exact"] >= 0.995 and st["max"] <= 1, st


@pytest.mark.parametrize("mode", ["bf16x3", "fp16x2"])
def test_depth_leg_bf16x3_4k_base_one_frame(R, mode):
    """configs[3]'s depth leg (DA-V2-Base at 3840x2160) in the split-bf16 mode, same bar."""
    from test_hip_depth_e2e import _plane_stats, _record, _stock_u8_planes
    from visiondepth3d_amd import synth
    from visiondepth3d_amd.depth import DepthPipe
    H, W = 2160, 3840
    frames = torch.from_numpy(np.stack([synth.synth_frame(0, H, W)[0]])).cuda()
    exp_u8, exp_pred = _stock_u8_planes("depth-anything-v2-base", frames)
    pipe = DepthPipe("depth-anything-v2-base", device="cuda", dtype=torch.float32, renderer=R, gemm=mode)
    pred = pipe.infer_bgr_u8(frames, raw=True)
    st = _plane_stats(R.depth_handoff(pred, H, W), exp_u8)
    st["pred_max_err_of_range"] = float((pred - exp_pred).abs().max()) / float(exp_pred.max() - exp_pred.min())
    _record(mode + "_4k_base", st)
    assert st["pred_max_err_of_range"] < 1e-4, st
    assert st["exact"] >= 0.995 and st["max"] <= 1, st
