"""Frame-interpolation network behind ``upscale.run_rife`` (core/merged_pipeline.py:33-60,204-218).

The reference feeds two BGR frames, concatenated along the channel axis and scaled to [0, 1], to an ONNX Runtime session
(``weights/RIFE_fp32.onnx``: input ``[N, 6, H, W]`` float32, output ``[N, 3, H, W]``, the frame half-way between the two) and keeps
its glue in NumPy.  The ONNX file is not part of the reference tree (``weights/WEIGHTS_README_PLACEHOLDER.md`` points to a download)
and there is no network here, so -- like the depth and up-scale networks -- the ARCHITECTURE is built from its published definition with
deterministic synthetic weights: RIFE v4 "IFNet HDv3" (hzwer/Practical-RIFE, MIT licence; restated from the paper "Real-Time
Intermediate Flow Estimation for Video Frame Interpolation", Huang et al., ECCV 2022, and the public model definition -- no code of it
is in /root/reference).  Three coarse-to-fine IFBlocks (scales 4, 2, 1; 90 channels) estimate the two intermediate flows and a fusion
mask; the frames are back-warped by ``grid_sample`` and blended.  **Parity unpinned** w.r.t. the reference's checkpoint (absent);
``tests/test_hip_upscale.py`` checks the GPU run of ``run_rife`` (HIP glue + this module on PyTorch-ROCm) against a CPU float32 run of
the same module, and ``RifeNet.load_state_dict`` accepts a Practical-RIFE ``flownet.pkl`` state dict (``block0..2`` keys) when one exists.

PyTorch-ROCm executes the convolutions (MIOpen); hand-written HIP is the glue on both sides (``vd3d_rife_preprocess`` /
``vd3d_rife_postprocess``), like the reference's split between its session and its NumPy lines.
"""
from __future__ import annotations

import zlib

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


def _conv(cin, cout, k=3, stride=1, pad=1):
    return nn.Sequential(nn.Conv2d(cin, cout, k, stride, pad, bias=True), nn.PReLU(cout))


class IFBlock(nn.Module):
    """One refinement block: two stride-2 convolutions, four residual pairs of 3x3 convolutions, two transposed-convolution heads
    (flow: 4 channels, mask: 1 channel).  ``scale``: the block works at 1/scale of the frame size."""

    def __init__(self, in_planes: int, c: int = 90):
        super().__init__()
        self.conv0 = nn.Sequential(_conv(in_planes, c // 2, 3, 2, 1), _conv(c // 2, c, 3, 2, 1))
        self.convblock0 = nn.Sequential(_conv(c, c), _conv(c, c))
        self.convblock1 = nn.Sequential(_conv(c, c), _conv(c, c))
        self.convblock2 = nn.Sequential(_conv(c, c), _conv(c, c))
        self.convblock3 = nn.Sequential(_conv(c, c), _conv(c, c))
        self.conv1 = nn.Sequential(nn.ConvTranspose2d(c, c // 2, 4, 2, 1), nn.PReLU(c // 2), nn.ConvTranspose2d(c // 2, 4, 4, 2, 1))
        self.conv2 = nn.Sequential(nn.ConvTranspose2d(c, c // 2, 4, 2, 1), nn.PReLU(c // 2), nn.ConvTranspose2d(c // 2, 1, 4, 2, 1))

    def forward(self, x, flow, scale: float = 1.0):
        x = F.interpolate(x, scale_factor=1.0 / scale, mode="bilinear", align_corners=False, recompute_scale_factor=False)
        flow = F.interpolate(flow, scale_factor=1.0 / scale, mode="bilinear", align_corners=False, recompute_scale_factor=False) * (1.0 / scale)
        feat = self.conv0(torch.cat((x, flow), 1))
        feat = self.convblock0(feat) + feat
        feat = self.convblock1(feat) + feat
        feat = self.convblock2(feat) + feat
        feat = self.convblock3(feat) + feat
        flow = self.conv1(feat)
        mask = self.conv2(feat)
        flow = F.interpolate(flow, scale_factor=scale, mode="bilinear", align_corners=False, recompute_scale_factor=False) * scale
        mask = F.interpolate(mask, scale_factor=scale, mode="bilinear", align_corners=False, recompute_scale_factor=False)
        return flow, mask


def backwarp(img: torch.Tensor, flow: torch.Tensor) -> torch.Tensor:
    """Sample ``img`` at ``pixel + flow`` (bilinear, border padding, align_corners=True): the ``warp`` of the published model."""
    n, _, h, w = img.shape
    xs = torch.linspace(-1.0, 1.0, w, device=img.device, dtype=img.dtype).view(1, 1, 1, w).expand(n, -1, h, -1)
    ys = torch.linspace(-1.0, 1.0, h, device=img.device, dtype=img.dtype).view(1, 1, h, 1).expand(n, -1, -1, w)
    grid = torch.cat((xs, ys), 1)
    fl = torch.cat((flow[:, 0:1] / ((w - 1.0) / 2.0), flow[:, 1:2] / ((h - 1.0) / 2.0)), 1)
    g = (grid + fl).permute(0, 2, 3, 1)
    return F.grid_sample(img, g, mode="bilinear", padding_mode="border", align_corners=True)


class RifeNet(nn.Module):
    """IFNet (HDv3, v4.0 form: the blocks see the two warped frames, the mask and the flow; the mid-point is implicit):
    ``forward(x [N,6,H,W] in [0,1]) -> [N,3,H,W]``, the frame half-way between x[:, :3] and x[:, 3:]."""

    def __init__(self, c: int = 90, scale_list=(4.0, 2.0, 1.0)):
        super().__init__()
        self.block0 = IFBlock(7 + 4, c)
        self.block1 = IFBlock(7 + 4, c)
        self.block2 = IFBlock(7 + 4, c)
        self.scale_list = tuple(float(s) for s in scale_list)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        n, _, h, w = x.shape
        ph, pw = (-h) % 32, (-w) % 32           # the coarsest block works at 1/16 of the frame: pad like the published inference script
        if ph or pw:
            x = F.pad(x, (0, pw, 0, ph))
        img0, img1 = x[:, :3], x[:, 3:6]
        flow = torch.zeros_like(x[:, :4])
        mask = torch.zeros_like(x[:, :1])
        w0, w1 = img0, img1
        for blk, sc in zip((self.block0, self.block1, self.block2), self.scale_list):
            f, m = blk(torch.cat((w0, w1, mask), 1), flow, scale=sc)        # 3 + 3 + 1 image channels, + 4 flow channels inside the block
            flow = flow + f
            mask = mask + m
            w0 = backwarp(img0, flow[:, :2])
            w1 = backwarp(img1, flow[:, 2:4])
        mk = torch.sigmoid(mask)
        out = w0 * mk + w1 * (1.0 - mk)
        return out[:, :, :h, :w]


@torch.no_grad()
def synthetic_weights_(model: nn.Module, seed: int = 0) -> None:
    """Deterministic, platform-stable weights (PCG64 keyed by the parameter name), small enough that the estimated flow stays within
    a few pixels: an interpolation that actually moves and blends content, not an identity."""
    for name, p in model.named_parameters():
        rng = np.random.Generator(np.random.PCG64((zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0xFFFFFFFF))
        if p.ndim >= 2:
            fan_in = int(np.prod(p.shape[1:]))
            a = float(np.sqrt(3.0 / max(fan_in, 1))) * 0.7
            v = rng.uniform(-a, a, size=tuple(p.shape)).astype(np.float32)
        elif name.endswith("bias"):
            v = rng.uniform(-0.02, 0.02, size=tuple(p.shape)).astype(np.float32)
        else:                                   # PReLU slopes
            v = np.full(tuple(p.shape), 0.25, np.float32)
        p.copy_(torch.from_numpy(v).to(p.dtype))


class RifeSession:
    """The callable ``upscale.run_rife`` takes as its session: ``[N,6,H,W] -> [N,3,H,W]`` on the device (float32 like ``RIFE_fp32.onnx``;
    channels_last memory for MIOpen)."""

    def __init__(self, device="cuda", dtype=torch.float32, net: RifeNet | None = None, seed: int = 0):
        self.device, self.dtype = torch.device(device), dtype
        if net is None:
            net = RifeNet()
            synthetic_weights_(net, seed)
        self.net = net.eval().to(self.device, dtype)
        if self.device.type == "cuda":
            self.net = self.net.to(memory_format=torch.channels_last)

    @torch.no_grad()
    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        x = x.to(self.device, self.dtype)
        if self.device.type == "cuda":
            x = x.contiguous(memory_format=torch.channels_last)
        return self.net(x)
