"""The up-scale stage: ``run_esrgan`` of core/merged_pipeline.py:237-284 with the frame resident in HBM.

The reference runs one of three Real-ESRGAN ONNX exports (VisionDepth3D.py:1094-1098) through onnxruntime, with numpy / cv2 glue
around it.  Here:

* the glue is HIP (``vd3d_esr_preprocess`` / ``vd3d_esr_postprocess`` / ``vd3d_resize_cubic_u8`` / ``vd3d_resize_area_u8`` /
  ``vd3d_add_weighted_u8``): the uint8 frame never leaves the GPU between the renderer and the encoder;
* the network is the published architecture of the model behind each export -- third-party code that is NOT in /root/reference
  (xinntao/Real-ESRGAN, BasicSR): ``SRVGGNetCompact`` (realesr-general-x4v3: 32 convs, realesr-animevideov3: 16 convs) and ``RRDBNet``
  (RealESRGAN_x4plus: 23 RRDB blocks) -- on PyTorch-ROCm in channels_last, fp16 like the reference's ``*_fp16.onnx`` files
  (float32 / bf16 selectable).  Weights come from the public ``.pth`` state dicts (``params_ema`` / ``params``) or straight from the
  reference's ONNX files (``load_onnx_initializers``: a minimal protobuf reader, no onnx package needed).

``Upscaler.run_esrgan`` keeps the reference's argument names and every step of its body, including the ones that look odd: the tiled
path writes each tile's top-left ``tile x tile`` corner of the 4x prediction into an INPUT-sized canvas (:266-284, ``out =
np.zeros_like(img)``), and the result is resized to ``frame * scale`` and then back to the ORIGINAL frame size with INTER_CUBIC
(:260-262) before ``target_size`` is applied.
"""
from __future__ import annotations

import struct

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

# model_name -> (architecture, keyword arguments); names are the reference's (VisionDepth3D.py:1094-1098)
MODEL_ZOO = {
    "RealESR_Gx4_fp16": ("srvgg", dict(num_feat=64, num_conv=32, upscale=4)),
    "RealESR_Animex4_fp16": ("srvgg", dict(num_feat=64, num_conv=16, upscale=4)),
    "RealESRGAN_x4_fp16": ("rrdb", dict(num_feat=64, num_block=23, num_grow_ch=32, scale=4)),
}
BLEND_ALPHA = {"LOW": 0.85, "MEDIUM": 0.5, "HIGH": 0.25}   # blend_images :231-236


class SRVGGNetCompact(nn.Module):
    """realesr-general-x4v3 / realesr-animevideov3: a plain conv + PReLU stack at input resolution, pixel-shuffle at the end, plus the
    nearest-neighbour up-sampled input.  Module names follow the public checkpoints (``body.N``)."""

    def __init__(self, num_in_ch=3, num_out_ch=3, num_feat=64, num_conv=16, upscale=4):
        super().__init__()
        self.upscale = upscale
        body = [nn.Conv2d(num_in_ch, num_feat, 3, 1, 1), nn.PReLU(num_parameters=num_feat)]
        for _ in range(num_conv):
            body += [nn.Conv2d(num_feat, num_feat, 3, 1, 1), nn.PReLU(num_parameters=num_feat)]
        body.append(nn.Conv2d(num_feat, num_out_ch * upscale * upscale, 3, 1, 1))
        self.body = nn.ModuleList(body)

    def forward(self, x):
        out = x
        for m in self.body:
            out = m(out)
        out = F.pixel_shuffle(out, self.upscale)
        return out + F.interpolate(x, scale_factor=self.upscale, mode="nearest")


class _RDB(nn.Module):
    def __init__(self, nf, gc):
        super().__init__()
        self.conv1 = nn.Conv2d(nf, gc, 3, 1, 1)
        self.conv2 = nn.Conv2d(nf + gc, gc, 3, 1, 1)
        self.conv3 = nn.Conv2d(nf + 2 * gc, gc, 3, 1, 1)
        self.conv4 = nn.Conv2d(nf + 3 * gc, gc, 3, 1, 1)
        self.conv5 = nn.Conv2d(nf + 4 * gc, nf, 3, 1, 1)

    def forward(self, x):
        a = F.leaky_relu(self.conv1(x), 0.2)
        b = F.leaky_relu(self.conv2(torch.cat((x, a), 1)), 0.2)
        c = F.leaky_relu(self.conv3(torch.cat((x, a, b), 1)), 0.2)
        d = F.leaky_relu(self.conv4(torch.cat((x, a, b, c), 1)), 0.2)
        return self.conv5(torch.cat((x, a, b, c, d), 1)) * 0.2 + x


class _RRDB(nn.Module):
    def __init__(self, nf, gc):
        super().__init__()
        self.rdb1, self.rdb2, self.rdb3 = _RDB(nf, gc), _RDB(nf, gc), _RDB(nf, gc)

    def forward(self, x):
        return self.rdb3(self.rdb2(self.rdb1(x))) * 0.2 + x


class RRDBNet(nn.Module):
    """RealESRGAN_x4plus: residual-in-residual dense blocks, two nearest x2 up-samplings.  Module names follow the public checkpoint."""

    def __init__(self, num_in_ch=3, num_out_ch=3, scale=4, num_feat=64, num_block=23, num_grow_ch=32):
        super().__init__()
        self.scale = scale
        if scale == 2:
            num_in_ch *= 4
        elif scale == 1:
            num_in_ch *= 16
        self.conv_first = nn.Conv2d(num_in_ch, num_feat, 3, 1, 1)
        self.body = nn.Sequential(*[_RRDB(num_feat, num_grow_ch) for _ in range(num_block)])
        self.conv_body = nn.Conv2d(num_feat, num_feat, 3, 1, 1)
        self.conv_up1 = nn.Conv2d(num_feat, num_feat, 3, 1, 1)
        self.conv_up2 = nn.Conv2d(num_feat, num_feat, 3, 1, 1)
        self.conv_hr = nn.Conv2d(num_feat, num_feat, 3, 1, 1)
        self.conv_last = nn.Conv2d(num_feat, num_out_ch, 3, 1, 1)

    def forward(self, x):
        if self.scale == 2:
            x = F.pixel_unshuffle(x, 2)
        elif self.scale == 1:
            x = F.pixel_unshuffle(x, 4)
        feat = self.conv_first(x)
        feat = feat + self.conv_body(self.body(feat))
        feat = F.leaky_relu(self.conv_up1(F.interpolate(feat, scale_factor=2, mode="nearest")), 0.2)
        feat = F.leaky_relu(self.conv_up2(F.interpolate(feat, scale_factor=2, mode="nearest")), 0.2)
        return self.conv_last(F.leaky_relu(self.conv_hr(feat), 0.2))


def conv_weight_fragments(weight: torch.Tensor) -> torch.Tensor:
    """A [64,64,3,3] convolution weight in the fragment order ``vd3d_conv3x3_c64_f16`` streams (include/vd3d.h): fp16
    [36 steps][2 channel tiles][64 lanes][8], element [(kh*3+kw)*4 + kc][t][l][j] = W[32t + (l & 31)][16kc + 8(l >> 5) + j][kh][kw]."""
    if tuple(weight.shape) != (64, 64, 3, 3):
        raise AssertionError("conv_weight_fragments takes a [64,64,3,3] weight")
    w = weight.detach().to(torch.float16).permute(2, 3, 0, 1).reshape(9, 2, 32, 4, 2, 8)      # [tap][t][i][kc][g][j]
    return w.permute(0, 3, 1, 4, 2, 5).reshape(36, 2, 64, 8).contiguous()                   # [tap][kc][t][g][i][j] -> [step][t][lane][j]


def head_weight_matrix(weight: torch.Tensor) -> torch.Tensor:
    """A [64,3,3,3] head convolution weight as the float32 [27,64] matrix ``vd3d_conv3x3_head_f16`` takes: row (kh*3 + kw)*3 + ic, column oc
    (the fp16 values of the checkpoint, widened)."""
    if tuple(weight.shape) != (64, 3, 3, 3):
        raise AssertionError("head_weight_matrix takes a [64,3,3,3] weight")
    return weight.detach().to(torch.float16).float().permute(2, 3, 1, 0).reshape(27, 64).contiguous()


def build_network(model_name: str = "RealESR_Gx4_fp16") -> nn.Module:
    arch, kw = MODEL_ZOO[model_name]
    return SRVGGNetCompact(**kw) if arch == "srvgg" else RRDBNet(**kw)


# ---- ONNX initialisers without the onnx package ---------------------------------------------------------------------------------
def _pb_fields(buf: memoryview):
    """Iterate (field number, wire type, value) over one protobuf message; length-delimited values come back as memoryviews."""
    i, n = 0, len(buf)
    while i < n:
        key = 0
        shift = 0
        while True:
            b = buf[i]; i += 1
            key |= (b & 0x7F) << shift
            shift += 7
            if not b & 0x80:
                break
        fno, wt = key >> 3, key & 7
        if wt == 0:
            v = 0
            shift = 0
            while True:
                b = buf[i]; i += 1
                v |= (b & 0x7F) << shift
                shift += 7
                if not b & 0x80:
                    break
            yield fno, wt, v
        elif wt == 1:
            yield fno, wt, buf[i:i + 8]; i += 8
        elif wt == 2:
            ln = 0
            shift = 0
            while True:
                b = buf[i]; i += 1
                ln |= (b & 0x7F) << shift
                shift += 7
                if not b & 0x80:
                    break
            yield fno, wt, buf[i:i + ln]; i += ln
        elif wt == 5:
            yield fno, wt, buf[i:i + 4]; i += 4
        else:
            raise ValueError(f"unsupported protobuf wire type {wt}")


def _pb_varints(buf: memoryview):
    """A packed repeated varint field."""
    out, i, n = [], 0, len(buf)
    while i < n:
        v = 0
        shift = 0
        while True:
            b = buf[i]; i += 1
            v |= (b & 0x7F) << shift
            shift += 7
            if not b & 0x80:
                break
        out.append(v)
    return out


_ONNX_DTYPES = {1: np.float32, 10: np.float16, 11: np.float64, 7: np.int64, 6: np.int32}


def _onnx_tensor(buf: memoryview):
    dims, dtype, name, raw, floats = [], 1, "", None, []
    for fno, wt, v in _pb_fields(buf):
        if fno == 1:
            dims += _pb_varints(v) if wt == 2 else [v]
        elif fno == 2:
            dtype = v
        elif fno == 8:
            name = bytes(v).decode()
        elif fno == 9:
            raw = bytes(v)
        elif fno == 4:      # float_data, packed
            floats += list(np.frombuffer(bytes(v), np.float32)) if wt == 2 else [struct.unpack("<f", bytes(v))[0]]
    if dtype not in _ONNX_DTYPES:
        return name, None
    if raw is not None:
        arr = np.frombuffer(raw, _ONNX_DTYPES[dtype])
    elif floats:
        arr = np.asarray(floats, np.float32)
    else:
        return name, None
    return name, arr.reshape(dims) if dims else arr


def load_onnx_initializers(path: str):
    """(nodes, tensors) of an ONNX file: ``nodes`` = [(op_type, [input names])] in graph order, ``tensors`` = {name: ndarray} for the
    float initialisers and the Constant nodes' tensors.  Reads ModelProto.graph (field 7) -> node (1) / initializer (5)."""
    with open(path, "rb") as f:
        model = memoryview(f.read())
    graph = next((v for fno, wt, v in _pb_fields(model) if fno == 7 and wt == 2), None)
    if graph is None:
        raise ValueError(f"{path}: no GraphProto")
    nodes, tensors = [], {}
    for fno, wt, v in _pb_fields(graph):
        if fno == 5 and wt == 2:
            name, arr = _onnx_tensor(v)
            if arr is not None:
                tensors[name] = arr
        elif fno == 1 and wt == 2:
            ins, outs, op = [], [], ""
            for f2, w2, v2 in _pb_fields(v):
                if f2 == 1:
                    ins.append(bytes(v2).decode())
                elif f2 == 2:
                    outs.append(bytes(v2).decode())
                elif f2 == 4:
                    op = bytes(v2).decode()
                elif f2 == 5 and op == "Constant":     # AttributeProto: t = field 5
                    for f3, w3, v3 in _pb_fields(v2):
                        if f3 == 5 and w3 == 2 and outs:
                            _, arr = _onnx_tensor(v3)
                            if arr is not None:
                                tensors[outs[0]] = arr
            nodes.append((op, ins))
    return nodes, tensors


def _load_from_onnx(net: nn.Module, path: str) -> None:
    """Conv / PRelu nodes in graph order fill the network's Conv2d / PReLU modules in definition order (a traced export keeps that
    order); shapes are checked."""
    nodes, tensors = load_onnx_initializers(path)
    convs = [m for m in net.modules() if isinstance(m, nn.Conv2d)]
    prelus = [m for m in net.modules() if isinstance(m, nn.PReLU)]
    ci = pi = 0
    with torch.no_grad():
        for op, ins in nodes:
            if op == "Conv":
                if ci >= len(convs):
                    raise ValueError(f"{path}: more Conv nodes than the architecture has")
                w = torch.from_numpy(np.array(tensors[ins[1]], np.float32))
                if tuple(w.shape) != tuple(convs[ci].weight.shape):
                    raise ValueError(f"{path}: Conv #{ci} weight {tuple(w.shape)} != {tuple(convs[ci].weight.shape)}")
                convs[ci].weight.copy_(w)
                if len(ins) > 2 and ins[2] in tensors:
                    convs[ci].bias.copy_(torch.from_numpy(np.array(tensors[ins[2]], np.float32)))
                else:
                    convs[ci].bias.zero_()
                ci += 1
            elif op == "PRelu":
                if pi >= len(prelus):
                    raise ValueError(f"{path}: more PRelu nodes than the architecture has")
                prelus[pi].weight.copy_(torch.from_numpy(np.array(tensors[ins[1]], np.float32)).reshape(-1))
                pi += 1
    if ci != len(convs) or pi != len(prelus):
        raise ValueError(f"{path}: {ci} Conv / {pi} PRelu nodes, the architecture has {len(convs)} / {len(prelus)}")


def run_rife(renderer, session, frame1, frame2, multiplier, dtype=torch.float32):
    """run_rife (core/merged_pipeline.py:204-218) with the frames resident in HBM.  ``session``: the interpolation network as a callable
    ``[N,6,H,W] -> [N,3,H,W]`` -- ``visiondepth3d_amd.rife.RifeSession`` (IFNet HDv3 on PyTorch-ROCm; the reference's ``RIFE_fp32.onnx`` is not
    in /root/reference) or any other; ``None`` returns ``[]`` like the reference without a loaded model (:205-206).  Returns
    ``multiplier - 1`` uint8 [H,W,3] tensors, like the reference's list of frames (all equal: the reference repeats ONE mid-point input)."""
    if session is None:
        return []
    f1 = (frame1 if torch.is_tensor(frame1) else torch.from_numpy(np.ascontiguousarray(frame1))).to(renderer.device).contiguous()
    f2 = (frame2 if torch.is_tensor(frame2) else torch.from_numpy(np.ascontiguousarray(frame2))).to(renderer.device).contiguous()
    if f1.dtype != torch.uint8 or f1.shape != f2.shape or f1.dim() != 3 or f1.shape[2] != 3:
        raise AssertionError("run_rife takes two uint8 [H,W,3] frames of one size")
    from . import _lib
    from ._abi import DT_BF16, DT_F16, DT_F32
    h, w = int(f1.shape[0]), int(f1.shape[1])
    x = torch.empty((1, 6, h, w), dtype=dtype, device=renderer.device)
    renderer._enter(f1, f2, x)
    _lib.check(renderer._L.vd3d_rife_preprocess(renderer._ctx, {torch.float32: DT_F32, torch.bfloat16: DT_BF16, torch.float16: DT_F16}[dtype],
                                                f1.data_ptr(), f2.data_ptr(), h, w, 0, x.data_ptr()))
    renderer.ordered_after()
    with torch.no_grad():
        pred = session(x.repeat(int(multiplier) - 1, 1, 1, 1)).float().contiguous()      # np.repeat(tensor, multiplier - 1, axis=0) :212
    outs = []
    for i in range(pred.shape[0]):
        o = torch.empty((h, w, 3), dtype=torch.uint8, device=renderer.device)
        renderer._enter(pred, o)
        _lib.check(renderer._L.vd3d_rife_postprocess(renderer._ctx, pred[i].data_ptr(), h, w, 0, o.data_ptr()))
        outs.append(o)
    return outs


class Upscaler:
    """One Real-ESRGAN network + the HIP glue.  ``renderer`` is a ``visiondepth3d_amd.render_3d.Renderer`` (its context and stream)."""

    def __init__(self, renderer, model_name: str = "RealESR_Gx4_fp16", net: nn.Module | None = None, dtype=torch.float16,
                 hip_body: bool = True):
        """``hip_body``: run the 64 -> 64 body layers of the compact (SRVGG) networks through the hand-written matrix-core kernel
        (``vd3d_conv3x3_c64_f16``: conv + bias + PReLU in one launch), and -- since round 3 -- the 3 -> 64 head, the 64 -> 3 r^2 tail convolution
        (same kernel, zero-padded) and pixel-shuffle + nearest add through HIP as well: no library call is left in the compact networks."""
        self.renderer = renderer
        self.device = renderer.device
        self.model_name = model_name
        self.dtype = dtype
        self.net = (net if net is not None else build_network(model_name)).to(self.device, dtype).eval()
        if self.device.type == "cuda":
            self.net = self.net.to(memory_format=torch.channels_last)
        self._body = None
        # the all-HIP path is built for the reference's compact networks: 3 input channels, 64 features, a tail of <= 64 channels,
        # x2 / x4 (vd3d_esr_tail_f32); anything else a caller hands in runs through the module graph
        if (hip_body and self.device.type == "cuda" and dtype == torch.float16 and isinstance(self.net, SRVGGNetCompact)
                and self.net.body[0].out_channels == 64 and self.net.body[0].in_channels == 3
                and int(getattr(self.net, "upscale", 0)) in (2, 4) and self.net.body[-1].out_channels <= 64):
            self._body = self._prepare_body()

    def _prepare_body(self):
        """Device-resident operands of every layer: head (weight matrix, bias, slope), body [(weight fragments, bias, PReLU slope)], tail (weight
        fragments and bias zero-padded from 3 r^2 to 64 output channels: the tail convolution runs on the body's matrix-core kernel)."""
        mods = list(self.net.body)
        dev = self.device
        self._head = (head_weight_matrix(mods[0].weight).to(dev), mods[0].bias.detach().float().contiguous().to(dev),
                      mods[1].weight.detach().float().contiguous().to(dev))
        layers = []
        for i in range(2, len(mods) - 1, 2):      # body[0..1] = head conv + PReLU, body[-1] = tail conv
            conv, act = mods[i], mods[i + 1]
            layers.append((conv_weight_fragments(conv.weight).to(dev), conv.bias.detach().float().contiguous().to(dev),
                           act.weight.detach().float().contiguous().to(dev)))
        tail = mods[-1]
        oc = tail.out_channels                    # 3 r^2: 48 (x4) or 12 (x2)
        wt = torch.zeros((64, 64, 3, 3), dtype=tail.weight.dtype)
        wt[:oc] = tail.weight.detach().cpu()
        bt = torch.zeros(64, dtype=torch.float32)
        bt[:oc] = tail.bias.detach().float().cpu()
        self._tail = (conv_weight_fragments(wt).to(dev), bt.to(dev))
        return layers

    def _forward(self, x: torch.Tensor) -> torch.Tensor:
        """The network on ``x`` ([1,3,H,W], channels_last).  Library path: the module's own dtype; HIP path (compact networks, fp16): every
        layer hand-written -- head (``vd3d_conv3x3_head_f16``), body and tail convolution (``vd3d_conv3x3_c64_f16``, matrix cores), pixel-shuffle
        + nearest add (``vd3d_esr_tail_f32``) -- and the result is already the float32 prediction."""
        if self._body is None:
            return self.net(x)
        R, net = self.renderer, self.net
        h = R.conv3x3_head(x, *self._head)
        spare = torch.empty_like(h, memory_format=torch.channels_last)
        for wf, b, sl in self._body:
            R.conv3x3_c64(h, wf, b, sl, out=spare)
            h, spare = spare, h
        R.conv3x3_c64(h, self._tail[0], self._tail[1], None, out=spare)
        out = R.esr_tail(spare, x, net.upscale)
        R.ordered_after()
        return out

    @classmethod
    def from_weights(cls, renderer, path: str, model_name: str = "RealESR_Gx4_fp16", dtype=torch.float16) -> "Upscaler":
        """``path``: a public ``.pth`` checkpoint (``params_ema`` / ``params`` / bare state dict) or one of the reference's ``.onnx`` files."""
        net = build_network(model_name)
        if path.lower().endswith(".onnx"):
            _load_from_onnx(net, path)
        else:
            sd = torch.load(path, map_location="cpu", weights_only=True)
            for key in ("params_ema", "params"):
                if isinstance(sd, dict) and key in sd:
                    sd = sd[key]
                    break
            net.load_state_dict(sd, strict=True)
        return cls(renderer, model_name, net, dtype)

    @property
    def scale(self) -> int:
        return 2 if "x2" in self.model_name.lower() else 4    # :259

    @torch.no_grad()
    def _infer(self, frame: torch.Tensor, y0=0, x0=0, h=None, w=None) -> torch.Tensor:
        """preprocess_esr -> network -> float32 [1,3,s*h,s*w] (the session output the reference post-processes)."""
        R = self.renderer
        x = R.esr_preprocess(frame, y0, x0, h, w, dtype=self.dtype, channels_last=True)
        R.ordered_after()
        return self._forward(x).float()

    def _esrgan_tiled(self, img: torch.Tensor, tile: int, pad: int) -> torch.Tensor:
        """_esrgan_tiled :266-284, quirk included: the canvas has the INPUT size, so each tile contributes the top-left corner of the
        centre region of its prediction."""
        R = self.renderer
        h, w = int(img.shape[0]), int(img.shape[1])
        out = torch.zeros_like(img)
        for y in range(0, h, tile):
            for x in range(0, w, tile):
                y0, x0 = max(0, y - pad), max(0, x - pad)
                y1, x1 = min(h, y + tile + pad), min(w, x + tile + pad)
                pred = self._infer(img, y0, x0, y1 - y0, x1 - x0)
                th, tw = min(tile, h - y), min(tile, w - x)
                R.esr_postprocess(pred, out=out, window=(y - y0, x - x0, th, tw), dst_yx=(y, x))
        return out

    def run_esrgan(self, frame, blend_mode="OFF", input_res_pct=100, model_name="RealESR_Gx4_fp16", target_size=None, tile=None, tile_pad=8):
        """core/merged_pipeline.py:237-264.  ``frame``: uint8 BGR [H,W,3] (tensor or array); returns a uint8 BGR tensor on the device."""
        R = self.renderer
        original = frame if torch.is_tensor(frame) else torch.from_numpy(np.ascontiguousarray(frame))
        original = original.to(self.device).contiguous()
        frame = original
        if input_res_pct != 100:
            h, w = int(frame.shape[0]), int(frame.shape[1])
            nh, nw = int(h * input_res_pct / 100), int(w * input_res_pct / 100)
            frame = R.resize_area_u8(frame, nh, nw) if input_res_pct < 100 else R.resize_cubic_u8(frame, nh, nw)
        if tile:
            upscaled = self._esrgan_tiled(frame, int(tile), int(tile_pad))
        else:
            upscaled = R.esr_postprocess(self._infer(frame))
        scale = 2 if "x2" in str(model_name).lower() else 4    # :259: from the CALL's model name (default literal 'RealESR_Gx4_fp16'), like the reference
        fh, fw = int(frame.shape[0]), int(frame.shape[1])
        upscaled = R.resize_cubic_u8(upscaled, fh * scale, fw * scale)
        upscaled = R.resize_cubic_u8(upscaled, int(original.shape[0]), int(original.shape[1]))
        if target_size:
            upscaled = R.resize_cubic_u8(upscaled, int(target_size[1]), int(target_size[0]))
        return self.blend_images(original, upscaled, blend_mode)

    def upscale(self, frame) -> torch.Tensor:
        """The network's own output at ``scale`` x the frame size (what ``run_esrgan`` would write if it did not resize back): the
        1080p -> 4K path of BASELINE configs[4]."""
        f = frame if torch.is_tensor(frame) else torch.from_numpy(np.ascontiguousarray(frame))
        return self.renderer.esr_postprocess(self._infer(f.to(self.device).contiguous()))

    def blend_images(self, original, upscaled, mode="OFF"):
        """blend_images :231-236 (``cv2.addWeighted`` needs equal shapes there too)."""
        if mode == "OFF":
            return upscaled
        alpha = BLEND_ALPHA.get(str(mode).upper(), 1.0)
        if tuple(original.shape) != tuple(upscaled.shape):
            raise AssertionError("blend_images: original and upscaled differ in size (cv2.addWeighted raises there)")
        return self.renderer.add_weighted_u8(upscaled, alpha, original, 1 - alpha, 0.0)
