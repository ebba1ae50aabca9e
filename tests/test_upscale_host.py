"""CPU tests of the up-scale stage and of the uint8 INTER_CUBIC resize of a24: the oracle's restatements against an
independent float implementation, the network definitions, and the two weight loaders (public .pth state dicts, the reference's ONNX
files through the built-in protobuf reader)."""
import struct

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.nn.functional as F  # noqa: E402


@pytest.mark.parametrize("shape,dsize", [((37, 53), (74, 106)), ((37, 53), (91, 140)), ((64, 48), (48, 36)), ((30, 40, 3), (120, 160)),
                                         ((50, 70, 3), (20, 33)), ((5, 4), (17, 9)), ((1, 9), (3, 27))])
def test_oracle_cubic_u8_matches_aten_bicubic_within_one_level(oracle, shape, dsize):
    """cv2 is absent, so OpenCV's fixed-point INTER_CUBIC (restated in oracle/) is cross-checked against ATen's float bicubic, which
    uses the same kernel (A = -0.75), the same half-pixel coordinate map and replicate borders: they may differ by the 11-bit
    coefficient quantisation only -- at most one level after rounding."""
    rng = np.random.default_rng(11)
    a = rng.integers(0, 256, shape, dtype=np.uint8)
    dh, dw = dsize
    got = oracle.resize_cubic_u8(a, dh, dw).astype(np.int32)
    t = torch.from_numpy(a.astype(np.float32))
    t = t[None, None] if t.dim() == 2 else t.permute(2, 0, 1)[None]
    ref = F.interpolate(t, size=(dh, dw), mode="bicubic", align_corners=False)
    ref = ref[0, 0] if a.ndim == 2 else ref[0].permute(1, 2, 0)
    ref = ref.numpy()
    want = np.clip(np.rint(ref), 0, 255).astype(np.int32)
    d = np.abs(got - want)
    # a difference of one level is allowed only where the float value sits near a rounding boundary or the coefficients' quantisation
    # error (<= 4 * 255 / 4096 per axis) can move it across
    assert d.max() <= 1, d.max()
    assert np.all(np.abs(got - np.clip(ref, 0, 255)) <= 0.5 + 0.51)


def test_oracle_cubic_u8_identity_constant_and_monotone(oracle):
    rng = np.random.default_rng(1)
    a = rng.integers(0, 256, (19, 23, 3), dtype=np.uint8)
    assert np.array_equal(oracle.resize_cubic_u8(a, 19, 23), a)                       # cv::resize copies at equal size
    c = np.full((9, 11), 173, np.uint8)
    assert np.all(oracle.resize_cubic_u8(c, 40, 31) == 173)                           # coefficients sum to 2048
    ramp = np.tile(np.arange(0, 200, 4, dtype=np.uint8), (6, 1))
    up = oracle.resize_cubic_u8(ramp, 6, 200).astype(int)
    assert np.all(np.diff(up[0][8:-8]) >= 0)


def test_esr_pre_post_are_the_numpy_one_liners(oracle):
    rng = np.random.default_rng(2)
    bgr = rng.integers(0, 256, (13, 17, 3), dtype=np.uint8)
    x = oracle.esr_pre(bgr)
    want = np.transpose(bgr[..., ::-1].astype(np.float32) / 255.0, (2, 0, 1))          # preprocess_esr :219-223
    assert np.array_equal(x, want.astype(np.float32))
    t = (rng.standard_normal((3, 13, 17)) * 0.7 + 0.5).astype(np.float32)
    got = oracle.esr_post(t)
    want = (np.clip(np.transpose(t, (1, 2, 0)), 0, 1) * 255.0).astype(np.uint8)[..., ::-1]   # postprocess_esr :225-229
    assert np.array_equal(got, want)
    assert np.array_equal(oracle.esr_post(oracle.esr_pre(bgr)), bgr)                  # u8 -> /255 -> *255 truncation round trip


def test_add_weighted_rounding(oracle):
    a = np.arange(256, dtype=np.uint8)
    b = a[::-1].copy()
    for alpha in (0.85, 0.5, 0.25):
        got = oracle.add_weighted_u8(a, alpha, b, 1 - alpha).astype(int)
        ref = a.astype(np.float64) * np.float32(alpha) + b.astype(np.float64) * np.float32(1 - alpha)
        assert np.all(np.abs(got - ref) <= 0.5 + 1e-4)
    assert np.array_equal(oracle.add_weighted_u8(a, 0.5, a, 0.5), a)


def test_networks_shapes_and_parameter_counts():
    from visiondepth3d_amd import upscale as U
    counts = {}
    for name in U.MODEL_ZOO:
        net = U.build_network(name).eval()
        counts[name] = sum(p.numel() for p in net.parameters())
        with torch.no_grad():
            y = net(torch.rand(1, 3, 12, 20))
        assert tuple(y.shape) == (1, 3, 48, 80)
    # the published sizes of realesr-general-x4v3, realesr-animevideov3 and RealESRGAN_x4plus
    assert counts == {"RealESR_Gx4_fp16": 1213296, "RealESR_Animex4_fp16": 621424, "RealESRGAN_x4_fp16": 16697987}


def test_srvgg_state_dict_keys_follow_the_public_checkpoint():
    from visiondepth3d_amd import upscale as U
    sd = U.build_network("RealESR_Animex4_fp16").state_dict()
    assert list(sd)[:3] == ["body.0.weight", "body.0.bias", "body.1.weight"]
    assert tuple(sd["body.34.weight"].shape) == (48, 64, 3, 3)
    sd = U.build_network("RealESRGAN_x4_fp16").state_dict()
    for k in ("conv_first.weight", "body.0.rdb1.conv1.weight", "body.22.rdb3.conv5.bias", "conv_body.weight", "conv_up1.weight",
              "conv_up2.weight", "conv_hr.weight", "conv_last.bias"):
        assert k in sd


# ---- a tiny ONNX writer (test-side only) to round-trip the built-in reader ------------------------------------------------------------
def _vi(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _ld(fno, payload):
    return _vi((fno << 3) | 2) + _vi(len(payload)) + payload


def _tensor(name, arr, packed_dims=True):
    dt = {np.dtype(np.float32): 1, np.dtype(np.float16): 10}[arr.dtype]
    dims = b"".join(_vi(d) for d in arr.shape)
    body = _ld(1, dims) if packed_dims else b"".join(_vi((1 << 3) | 0) + _vi(d) for d in arr.shape)
    return body + _vi((2 << 3) | 0) + _vi(dt) + _ld(8, name.encode()) + _ld(9, arr.tobytes())


def _node(op, ins, outs):
    return b"".join(_ld(1, i.encode()) for i in ins) + b"".join(_ld(2, o.encode()) for o in outs) + _ld(4, op.encode())


def _onnx_bytes(net, fp16, anonymous):
    import torch.nn as nn
    nodes, inits, k = [], [], 0
    prev = "input"
    for m in net.modules():
        if isinstance(m, nn.Conv2d):
            wn, bn = (f"onnx::Conv_{k}", f"onnx::Conv_{k + 1}") if anonymous else (f"c{k}.weight", f"c{k}.bias")
            w, b = m.weight.detach().numpy(), m.bias.detach().numpy()
            if fp16:
                w, b = w.astype(np.float16), b.astype(np.float16)
            inits += [_tensor(wn, w), _tensor(bn, b, packed_dims=False)]
            nodes.append(_node("Conv", [prev, wn, bn], [f"t{k}"]))
            nodes.append(_node("Cast", [f"t{k}"], [f"t{k}c"]))      # noise the reader has to skip
            prev = f"t{k}c"
            k += 2
        elif isinstance(m, nn.PReLU):
            sn = f"slope{k}"
            inits.append(_tensor(sn, m.weight.detach().numpy().reshape(-1, 1, 1).astype(np.float16 if fp16 else np.float32)))
            nodes.append(_node("PRelu", [prev, sn], [f"t{k}"]))
            prev = f"t{k}"
            k += 1
    graph = b"".join(_ld(1, n) for n in nodes) + _ld(2, b"g") + b"".join(_ld(5, t) for t in inits)
    return _vi((1 << 3) | 0) + _vi(8) + _ld(2, b"test") + _ld(7, graph)


@pytest.mark.parametrize("name,fp16,anonymous", [("RealESR_Animex4_fp16", True, False), ("RealESR_Animex4_fp16", False, True)])
def test_onnx_initialisers_round_trip(tmp_path, name, fp16, anonymous):
    from visiondepth3d_amd import upscale as U
    torch.manual_seed(5)
    src = U.build_network(name)
    with torch.no_grad():
        for p in src.parameters():
            p.copy_(torch.randn_like(p) * 0.05)
    path = tmp_path / "m.onnx"
    path.write_bytes(_onnx_bytes(src, fp16, anonymous))
    dst = U.build_network(name)
    U._load_from_onnx(dst, str(path))
    for (k, a), (_, b) in zip(src.state_dict().items(), dst.state_dict().items()):
        want = a.half().float() if fp16 else a
        assert torch.equal(want, b), k
    # a file of another architecture is refused
    other = U.build_network("RealESR_Gx4_fp16")
    with pytest.raises(ValueError):
        U._load_from_onnx(other, str(path))


def test_pth_checkpoint_round_trip(tmp_path):
    from visiondepth3d_amd import upscale as U

    class FakeRenderer:
        device = torch.device("cpu")

    src = U.build_network("RealESR_Animex4_fp16")
    torch.save({"params_ema": src.state_dict()}, tmp_path / "w.pth")
    up = U.Upscaler.from_weights(FakeRenderer(), str(tmp_path / "w.pth"), "RealESR_Animex4_fp16", dtype=torch.float32)
    for (k, a), (_, b) in zip(src.state_dict().items(), up.net.state_dict().items()):
        assert torch.equal(a, b), k
    assert up.scale == 4


def test_rife_network_on_cpu_shapes_padding_and_determinism():
    """The interpolation network behind run_rife (visiondepth3d_amd/rife.py, IFNet HDv3 with synthetic weights) on the CPU: the
    [N,6,H,W] -> [N,3,H,W] session contract of core/merged_pipeline.py:204-218, sizes that are not multiples of 32, determinism, the
    state-dict key names of the published checkpoints, and an output that is neither of the inputs nor their plain average."""
    import numpy as np
    import torch
    from visiondepth3d_amd import synth
    from visiondepth3d_amd.rife import RifeNet, RifeSession
    s = RifeSession(device="cpu")
    keys = set(s.net.state_dict())
    for k in ("block0.conv0.0.0.weight", "block0.conv0.0.1.weight", "block1.convblock3.1.0.bias", "block2.conv1.0.weight", "block2.conv1.1.weight",
              "block2.conv2.2.weight"):
        assert k in keys, k
    assert s.net.state_dict()["block0.conv0.0.0.weight"].shape == (45, 11, 3, 3)      # 3 + 3 + 1 image channels + 4 flow channels -> c / 2
    assert s.net.state_dict()["block2.conv1.2.weight"].shape == (45, 4, 4, 4)         # transposed convolution to the 4 flow channels
    f1, _ = synth.synth_frame(1, 70, 100)
    f2, _ = synth.synth_frame(2, 70, 100)
    x = torch.from_numpy(np.concatenate((f1.astype(np.float32) / 255.0, f2.astype(np.float32) / 255.0), axis=2)).permute(2, 0, 1)[None]
    y = s(x.repeat(2, 1, 1, 1))
    assert tuple(y.shape) == (2, 3, 70, 100) and torch.allclose(y[0], y[1], atol=1e-5, rtol=0)   # (oneDNN blocks differently per batch size)
    assert torch.allclose(y[0], RifeSession(device="cpu")(x)[0], atol=1e-5, rtol=0)     # same weights, same result
    assert 0.0 < float(y.min()) and float(y.max()) < 1.0
    mid = (x[:, :3] + x[:, 3:]) / 2
    assert 1e-3 < float((y[:1] - mid).abs().mean()) < 0.05                              # a warp + blend, not a plain average ...
    assert float((y[:1] - x[:, :3]).abs().mean()) > float((y[:1] - mid).abs().mean())  # ... and closer to the mid-point than to a frame
    n = RifeNet()
    n.load_state_dict(s.net.state_dict())                                               # round trip like a checkpoint


def test_head_weight_matrix_and_tail_padding_layouts():
    """The operand layouts the hand-written head / tail layers take (include/vd3d.h), pinned on the CPU against ATen: the head's [27,64]
    matrix is the convolution weight in (kh, kw, ic) x oc order; the tail convolution zero-padded to 64 output channels followed by the
    kernel's pixel-shuffle index rule (channel c r^2 + i r + j of pixel (h, w) -> out[c][h r + i][w r + j]) and the nearest-neighbour add is the
    network's own tail."""
    from visiondepth3d_amd.upscale import SRVGGNetCompact, conv_weight_fragments, head_weight_matrix
    torch.manual_seed(5)
    x = torch.rand(1, 3, 9, 11)
    w = torch.randn(64, 3, 3, 3) * 0.2
    m = head_weight_matrix(w)                                   # float32 [27, 64] of the fp16-rounded weights
    assert m.dtype == torch.float32 and tuple(m.shape) == (27, 64)
    cols = F.unfold(x, 3, padding=1)                            # [1, ic*kh*kw, L] in (ic, kh, kw) order
    cols = cols.view(1, 3, 3, 3, -1).permute(0, 2, 3, 1, 4).reshape(1, 27, -1)   # -> (kh, kw, ic)
    got = torch.einsum("tl,to->ol", cols[0], m).view(1, 64, 9, 11)
    assert torch.allclose(got, F.conv2d(x, w.half().float(), padding=1), atol=1e-5)
    for r in (4, 2):
        net = SRVGGNetCompact(num_conv=1, upscale=r).eval()
        tail = net.body[-1]
        oc = tail.out_channels
        assert oc == 3 * r * r
        wt = torch.zeros(64, 64, 3, 3)
        wt[:oc] = tail.weight.detach()
        assert tuple(conv_weight_fragments(wt).shape) == (36, 2, 64, 8)
        feat = torch.randn(1, 64, 6, 7)
        t = F.conv2d(feat, wt, torch.cat([tail.bias.detach(), torch.zeros(64 - oc)]), padding=1)      # what the padded MFMA layer computes
        assert float(t[:, oc:].abs().max()) == 0.0
        out = torch.empty(1, 3, 6 * r, 7 * r)
        xin = torch.rand(1, 3, 6, 7)
        for c in range(3):
            for i in range(r):
                for j in range(r):
                    out[0, c, i::r, j::r] = t[0, c * r * r + i * r + j] + xin[0, c]
        exp = F.pixel_shuffle(tail(feat), r) + F.interpolate(xin, scale_factor=r, mode="nearest")
        assert torch.allclose(out, exp, atol=1e-6)
