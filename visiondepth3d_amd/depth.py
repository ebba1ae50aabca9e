"""Depth-model callable (boundary B3, core/render_depth.py:1106-1119) on PyTorch-ROCm.

The reference wraps ``transformers.pipeline("depth-estimation")`` in
``pipe(images: list[PIL], inference_size=None) -> list[{"predicted_depth": Tensor}]``.  This module
keeps that protocol (``DepthPipe.__call__``) and adds a device-resident fast path
(``DepthPipe.infer_bgr_u8``) that consumes uint8 BGR frames already in HBM and returns depth planes in HBM,
so the 2D->3D stage can start without the reference's host round trip and 8-bit depth video on disk
(core/render_depth.py:1907-1935) -- while still reproducing that hand-off's per-frame min-max -> uint8
truncation (``depth_to_u8``, a24) so the DIBR stage sees the same quantised values.

MFMA work (DINOv2 GEMMs / attention, DPT convs) goes through PyTorch-ROCm (hipBLASLt / MIOpen / SDPA): per
BASELINE.json:north_star the depth net is NOT hand-written.

Precision: the reference loads ``AutoModelForDepthEstimation.from_pretrained(checkpoint)`` with no dtype, i.e. float32
(core/render_depth.py:758-759,823-824); float32 is the DEFAULT here (f32-input MFMA on gfx950) and bfloat16 is an opt-in
whose uint8 depth-plane deviation is measured by tests/test_hip_depth_e2e.py.

Checkpoints: ``DepthPipe.from_pretrained(dir)`` loads a local Hugging Face folder (config.json + model.safetensors +
preprocessor_config.json) exactly like the reference's local-model branch (:756-760) and then applies the fused-weight
rewrites.  No network and no checkpoints exist in the build environment, so benchmarks and tests use deterministic
synthetic weights (NumPy PCG64 keyed by parameter name) on the real architectures.
"""
from __future__ import annotations

import json
import os
import zlib

import numpy as np
import torch
import torch.nn.functional as F

# Architectures of the reference's model list (core/render_depth.py:686-712) that go through its Hugging Face branch
# (AutoModelForDepthEstimation, :756-760,823-824).  "da" = DepthAnythingForDepthEstimation on a DINOv2 backbone: Depth-Anything V1
# (LiheYoung/depth-anything-*-hf), V2 (depth-anything/Depth-Anything-V2-*-hf) and Distill-Any-Depth
# (xingyang1/Distill-Any-Depth-*-hf) share these three shapes and differ only in their weights.  "dpt" = DPTForDepthEstimation
# (MiDaS 3.0 family: Intel/dpt-large).  Marigold / DepthCrafter / ONNX entries are other pipelines (out of scope).
_DA_S = dict(arch="da", hidden=384, layers=12, heads=6, out_indices=[9, 10, 11, 12], neck=[48, 96, 192, 384], fusion=64, head=32)
_DA_B = dict(arch="da", hidden=768, layers=12, heads=12, out_indices=[9, 10, 11, 12], neck=[96, 192, 384, 768], fusion=128, head=32)
_DA_L = dict(arch="da", hidden=1024, layers=24, heads=16, out_indices=[21, 22, 23, 24], neck=[256, 512, 1024, 1024], fusion=256, head=32)
MODEL_ZOO = {
    "depth-anything-v2-small": _DA_S, "depth-anything-v2-base": _DA_B, "depth-anything-v2-large": _DA_L,
    "depth-anything-v1-small": _DA_S, "depth-anything-v1-base": _DA_B, "depth-anything-v1-large": _DA_L,
    "distill-any-depth-small": _DA_S, "distill-any-depth-large": _DA_L,
    "dpt-large": dict(arch="dpt", hidden=1024, layers=24, heads=16, out_indices=[5, 11, 17, 23], neck=[256, 512, 1024, 1024], fusion=256),
}
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)
# image-processor constants (preprocessor_config.json of the checkpoints): DA = DPTImageProcessor(size 518, keep_aspect_ratio,
# ensure_multiple_of 14, bicubic, ImageNet mean/std); DPT-Large = 384 x 384, no aspect keeping, mean = std = 0.5
PROCESSORS = {
    "da": dict(size=(518, 518), keep_aspect_ratio=True, multiple=14, mean=IMAGENET_MEAN, std=IMAGENET_STD),
    "dpt": dict(size=(384, 384), keep_aspect_ratio=False, multiple=1, mean=(0.5, 0.5, 0.5), std=(0.5, 0.5, 0.5)),
}


def build_config(name: str):
    from transformers import DepthAnythingConfig, Dinov2Config
    z = MODEL_ZOO[name]
    if z["arch"] == "dpt":
        from transformers import DPTConfig
        return DPTConfig(hidden_size=z["hidden"], num_hidden_layers=z["layers"], num_attention_heads=z["heads"],
                         intermediate_size=4 * z["hidden"], image_size=384, patch_size=16, backbone_out_indices=z["out_indices"],
                         neck_hidden_sizes=z["neck"], fusion_hidden_size=z["fusion"], readout_type="project")
    bc = Dinov2Config(hidden_size=z["hidden"], num_hidden_layers=z["layers"], num_attention_heads=z["heads"], patch_size=14,
                      image_size=518, out_indices=z["out_indices"], reshape_hidden_states=False, apply_layernorm=True)
    return DepthAnythingConfig(backbone_config=bc, reassemble_hidden_size=z["hidden"], neck_hidden_sizes=z["neck"],
                               fusion_hidden_size=z["fusion"],
                               head_hidden_size=z["head"], reassemble_factors=[4, 2, 1, 0.5], patch_size=14)


@torch.no_grad()
def synthetic_weights_(model: torch.nn.Module, seed: int = 0) -> None:
    """Deterministic, platform-stable weights: PCG64 keyed by crc32(parameter name) ^ seed."""
    for name, p in model.named_parameters():
        rng = np.random.Generator(np.random.PCG64((zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0xFFFFFFFF))
        if p.ndim >= 2 and "position_embeddings" not in name and "token" not in name:
            fan_in = int(np.prod(p.shape[1:]))
            a = float(np.sqrt(3.0 / max(fan_in, 1)))  # unit-gain uniform
            v = rng.uniform(-a, a, size=tuple(p.shape)).astype(np.float32)
        elif name.endswith("bias"):
            v = rng.uniform(-0.02, 0.02, size=tuple(p.shape)).astype(np.float32)
        elif "lambda" in name or "layer_scale" in name:
            v = np.full(tuple(p.shape), 0.2, np.float32)
        elif p.ndim == 1:
            v = (1.0 + rng.uniform(-0.05, 0.05, size=tuple(p.shape))).astype(np.float32)
        else:
            v = rng.uniform(-0.02, 0.02, size=tuple(p.shape)).astype(np.float32)
        p.copy_(torch.from_numpy(v).to(p.dtype))


def dpt_resize_target(h: int, w: int, size=518, multiple: int = 14, keep_aspect_ratio: bool = True):
    """DPTImageProcessor.get_resize_output_image_size (keep_aspect_ratio, ensure_multiple_of=14): 1080x1920 -> 518x924."""
    size_h, size_w = (size, size) if isinstance(size, int) else size
    sh, sw = size_h / h, size_w / w
    if keep_aspect_ratio:
        if abs(1 - sw) < abs(1 - sh):
            sh = sw
        else:
            sw = sh

    def snap(v):
        return max(int(round(v / multiple)) * multiple, multiple)

    return snap(sh * h), snap(sw * w)


def depth_to_u8(pred: torch.Tensor, invert: bool = False) -> torch.Tensor:
    """convert_depth_to_grayscale (core/render_depth.py:585-611) per frame, on device: min-max normalise,
    ``(norm*255).astype(uint8)`` truncation; invalid / flat frames -> zeros; optional ``255 - u8`` (:1914-1916)."""
    p = pred.float()
    flat = p.flatten(1)
    mn, mx = flat.min(dim=1).values, flat.max(dim=1).values
    bad = torch.isnan(mn) | torch.isnan(mx) | ((mx - mn) < 1e-6)
    norm = (p - mn[:, None, None]) / (mx - mn + 1e-6)[:, None, None]
    u8 = (norm * 255).clamp(0, 255).to(torch.uint8)
    u8 = torch.where(bad[:, None, None], torch.zeros_like(u8), u8)
    return 255 - u8 if invert else u8


_TUNABLE_DIR = None


def _tunable_scratch_dir() -> str:
    """One private scratch directory per process for TunableOp's results file (see DepthPipe._library_selection)."""
    global _TUNABLE_DIR
    if _TUNABLE_DIR is None:
        import atexit
        import glob
        import shutil
        import tempfile
        import time
        base = tempfile.gettempdir()
        for old in glob.glob(os.path.join(base, "vd3d_tunable_*")):   # what earlier processes of this user left behind (TunableOp writes at exit, after us)
            try:
                if os.stat(old).st_uid == os.getuid() and time.time() - os.stat(old).st_mtime > 86400:
                    shutil.rmtree(old, ignore_errors=True)
            except OSError:
                pass
        _TUNABLE_DIR = tempfile.mkdtemp(prefix="vd3d_tunable_")
        atexit.register(shutil.rmtree, _TUNABLE_DIR, True)
    return _TUNABLE_DIR


class DepthPipe:
    """``pipe(images, inference_size=None) -> [{"predicted_depth": Tensor[h, w]}]`` (reference protocol) plus a
    device-resident batch path."""

    def __init__(self, name: str = "depth-anything-v2-small", device="cuda", dtype=torch.float32, seed: int = 0,
                 channels_last: bool = True, renderer=None, fuse_backbone: bool = True, model=None, processor: dict | None = None,
                 tuned_gemm: bool = True, miopen_find: bool | None = None, gemm: str = "f32"):
        """``dtype``: float32 (the reference's precision, default) or bfloat16.
        ``gemm`` (float32 + ``renderer`` only; round 6): ``"f32"`` (default) -- the four linears of every transformer block are hipBLASLt's float32
        GEMMs; ``"bf16x3"`` -- OPT-IN: the library's own split-bf16 GEMM (``vd3d_gemm_x3``: every float32 operand exactly split into three bf16
        terms, six products per MAC on the bf16 matrix cores, float32 accumulation -- float32-faithful, see include/vd3d.h), with the exact GELU
        folded into fc1's epilogue, and the attention in the same arithmetic (``vd3d_attention_x3``: both products split-bf16, float32 online softmax).
        ``"fp16x2"`` -- OPT-IN: the same kernels with every operand as TWO fp16 terms (22 significant bits, round to nearest) and three products per MAC
        -- half the matrix work of bf16x3; weights pre-scaled per row, activations must stay below 65 504 (include/vd3d.h).  gfx950 has no TF32 and its float32-input MFMA runs at 1/16 of the bf16 rate, which caps the default mode.
        ``renderer``: a ``visiondepth3d_amd.render_3d.Renderer`` on the SAME stream as the network (default stream); when given the
        image-processor front end, the residual-add + LayerNorm pairs and the DPT up-samplings run as fused HIP launches.
        ``model`` / ``processor``: an already constructed Hugging Face depth model and its image-processor constants
        (``from_pretrained``); otherwise the architecture ``name`` is built from MODEL_ZOO with synthetic weights.
        Library selection (the GEMMs / convolutions stay library calls, north_star): ``tuned_gemm`` loads the committed hipBLASLt solution
        table ``tuned/gemm_gfx950.csv`` (PyTorch TunableOp, tuning OFF: a look-up per GEMM shape; written by tools/probe_net_tune.py on an
        MI355X; ignored when its library-version validators do not match; ``VD3D_TUNED_GEMM=0`` disables).  ``miopen_find``: MIOpen
        find mode (``torch.backends.cudnn.benchmark``) -- every convolution shape times its applicable solvers once, ~25 s at the
        first forward of a process, 4.5 % on the 4K float32 forward (profiles/r04_net_library_selection.md); opt-in, bench.py uses it.
        The flag is PyTorch's process-wide one: True / False set it, None (default) leaves it as the caller has it."""
        self.name, self.device, self.dtype = name, torch.device(device), dtype
        if gemm not in ("f32", "bf16x3", "fp16x2"):
            raise ValueError("gemm must be 'f32', 'bf16x3' or 'fp16x2'")
        if gemm != "f32" and (dtype != torch.float32 or renderer is None or torch.device(device).type != "cuda"):
            raise ValueError("gemm='bf16x3' / 'fp16x2' are modes of the float32 pipe on the GPU and need a renderer (the kernels live in libvd3d_hip.so)")
        self.gemm = gemm
        self.tuned_gemm = self.miopen_find = False
        self._flop_count = None
        if self.device.type == "cuda":
            self._library_selection(tuned_gemm, miopen_find)
        if dtype not in (torch.float32, torch.bfloat16):
            raise TypeError("DepthPipe runs in float32 (reference precision) or bfloat16")
        if model is None:
            cfg = build_config(name)
            if MODEL_ZOO[name]["arch"] == "dpt":
                from transformers import DPTForDepthEstimation as Net
            else:
                from transformers import DepthAnythingForDepthEstimation as Net
            model = Net(cfg).eval()
            synthetic_weights_(model, seed)
            processor = processor or PROCESSORS[MODEL_ZOO[name]["arch"]]
        self.arch = "da" if type(model).__name__ == "DepthAnythingForDepthEstimation" else "generic"
        self.proc = dict(PROCESSORS["da"] if processor is None else processor)
        self.model = model.eval().to(self.device, dtype)
        if channels_last and self.device.type == "cuda":
            self.model = self.model.to(memory_format=torch.channels_last)
        self.mean = torch.tensor(self.proc["mean"], device=self.device, dtype=torch.float32).view(1, 3, 1, 1)
        self.std = torch.tensor(self.proc["std"], device=self.device, dtype=torch.float32).view(1, 3, 1, 1)
        self.n_params = sum(p.numel() for p in self.model.parameters())
        self.renderer = renderer if (renderer is not None and self.device.type == "cuda") else None
        dino = self.arch == "da" and type(self.model.backbone).__name__ == "Dinov2Backbone"
        if self.gemm != "f32" and not (dino and fuse_backbone):
            raise NotImplementedError("gemm='bf16x3' / 'fp16x2' rewrite the fused DINOv2 backbone (Depth-Anything V1 / V2, Distill-Any-Depth with fuse_backbone=True); "
                                      "this model runs its stock float32 graph -- nothing is substituted silently")
        if dino:
            self._cache_position_embeddings()
            if fuse_backbone:
                self._fuse_backbone_layers()
            if self.renderer is not None:
                self._patch_dpt_upsampling()

    def _library_selection(self, tuned_gemm: bool, miopen_find):
        if miopen_find is not None:
            torch.backends.cudnn.benchmark = bool(miopen_find)
        self.miopen_find = bool(torch.backends.cudnn.benchmark)
        table = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuned", "gemm_gfx950.csv")
        if not tuned_gemm or os.environ.get("VD3D_TUNED_GEMM", "1") == "0" or not os.path.exists(table):
            return
        try:
            import tempfile
            import torch.cuda.tunable as tn
            if tn.is_enabled():       # the caller drives TunableOp itself (tools/probe_net_tune.py, PYTORCH_TUNABLEOP_ENABLED=1)
                return
            tn.enable(True)
            tn.tuning_enable(False)
            # never write into the package.  TunableOp rewrites its results file when the process exits and this build has no switch for that: the
            # file lives in a PRIVATE per-process directory (mkdtemp: mode 0700, unpredictable name -- no race between the ranks of --gpus N, no
            # symlink to clobber in a shared /tmp) that an atexit hook registered BEFORE TunableOp's own write... cannot rely on ordering,
            # so the hook removes what exists and the directory is also swept at the next start-up of the same user (stale entries older than a day)
            tn.set_filename(os.path.join(_tunable_scratch_dir(), "tunableop_results.csv"))
            self.tuned_gemm = bool(tn.read_file(table))
            if not self.tuned_gemm:   # other hipBLASLt / rocBLAS / PyTorch build: the solution indices mean nothing there
                tn.enable(False)
        except Exception:             # TunableOp missing in this build: the heuristic selection is the same math
            self.tuned_gemm = False

    @classmethod
    def from_pretrained(cls, path: str, device="cuda", dtype=torch.float32, **kw) -> "DepthPipe":
        """A local Hugging Face checkpoint folder (config.json + model.safetensors [+ preprocessor_config.json]) -- the reference's
        local-model branch, ``AutoModelForDepthEstimation.from_pretrained(checkpoint)`` + ``AutoProcessor.from_pretrained(checkpoint)``
        (core/render_depth.py:756-760).  DepthAnything / DINOv2 checkpoints (Depth-Anything V1 / V2, Distill-Any-Depth) then get the
        fused-weight rewrites (one QKV GEMM, LayerScale folded into the projections); any other depth architecture
        (DPT / MiDaS 3.0, ZoeDepth ...) runs its stock module graph behind the same protocol."""
        from transformers import AutoModelForDepthEstimation
        if not os.path.isdir(path):
            raise FileNotFoundError(f"{path}: not a local checkpoint folder (this build environment has no network)")
        model = AutoModelForDepthEstimation.from_pretrained(path, local_files_only=True)
        proc = dict(PROCESSORS["da" if type(model).__name__ == "DepthAnythingForDepthEstimation" else "dpt"])
        pc = os.path.join(path, "preprocessor_config.json")
        if os.path.exists(pc):
            with open(pc) as f:
                j = json.load(f)
            sz = j.get("size") or {}
            if isinstance(sz, dict) and "height" in sz:
                proc["size"] = (int(sz["height"]), int(sz["width"]))
            elif isinstance(sz, int):
                proc["size"] = (sz, sz)
            proc["keep_aspect_ratio"] = bool(j.get("keep_aspect_ratio", proc["keep_aspect_ratio"]))
            proc["multiple"] = int(j.get("ensure_multiple_of", proc["multiple"]))
            if j.get("image_mean") is not None:
                proc["mean"] = tuple(float(v) for v in j["image_mean"])
            if j.get("image_std") is not None:
                proc["std"] = tuple(float(v) for v in j["image_std"])
            if int(j.get("resample", 3)) != 3:
                raise NotImplementedError("image processors other than bicubic (PIL resample 3) are not built")
        return cls(os.path.basename(os.path.normpath(path)), device=device, dtype=dtype, model=model, processor=proc, **kw)

    def resize_target(self, h: int, w: int):
        return dpt_resize_target(h, w, self.proc["size"], self.proc["multiple"], self.proc["keep_aspect_ratio"])

    def _cache_position_embeddings(self):
        """DINOv2 re-interpolates its position embedding (bicubic 37x37 -> patch grid) on EVERY forward; for a fixed
        frame size the result is a constant, and ATen's bicubic kernel loops the 384-1024 channels serially per output
        pixel (measured 2.9 ms per forward on MI355X, 14 % of the step).  Memoise it per (height, width)."""
        emb = self.model.backbone.embeddings
        orig = emb.interpolate_pos_encoding
        cache = {}

        def cached(embeddings, height, width):
            key = (int(height), int(width), embeddings.dtype, embeddings.shape[1])
            if key not in cache:
                cache[key] = orig(embeddings, height, width).detach()
            return cache[key]

        emb.interpolate_pos_encoding = cached

    def _patch_dpt_upsampling(self):
        """The DPT neck / head between the library convolutions (transformers DepthAnythingReassembleStage / PreActResidualLayer /
        FeatureFusionLayer / DepthEstimationHead; the module graphs are otherwise reproduced verbatim):
          * the align_corners=True bilinear up-samplings run on vd3d_upsample_bilinear_nhwc;
          * float32 (the reference's precision): every convolution runs WITHOUT its bias pass and the glue between two convolutions is one
            HIP launch -- bias + ReLU in place behind a unit's first convolution; bias + unit input (+ the fusion layer's running state) behind
            its second one, which also writes the ReLU'd copy the next unit starts from; a convolution in front of an up-sampling hands its bias
            to the up-sampling kernel (interpolation weights sum to one); the fusion layer's 1x1 projection runs BEFORE its up-sampling (both
            are linear and per-pixel / per-channel: conv1x1(up(x)) == up(conv1x1(x)), a quarter of the pixels); behind the head's second
            convolution one launch does bias, ReLU, the 1x1 convolution to one channel, bias, ReLU and max_depth (vd3d_dpt_head_tail_f32);
          * the reassemble stage hands the token tensor to its 1x1 projection as a channels_last VIEW ([B,T,C] minus CLS is NHWC storage)
            instead of a NCHW copy that MIOpen converts back.
        Same function as the module graph up to float32 association (tests/test_hip_depth_e2e.py: 1e-4 of the range on the prediction,
        identical uint8 planes up to the stated bar)."""
        R = self.renderer
        f32 = self.dtype == torch.float32 and os.environ.get("VD3D_NECK_GLUE", "1") != "0"   # 0: A/B switch back to the module graph's own passes
        CL = torch.channels_last

        def up(x, size):
            if x.dtype in (torch.bfloat16, torch.float32) and x.shape[1] % 8 == 0 and x.is_contiguous(memory_format=CL):
                return R.upsample_bilinear(x, size)
            return F.interpolate(x, size=size, mode="bilinear", align_corners=True)

        conv_img = {}   # fp16x2 mode: packed weights per convolution module (None = shape not built: the library convolution stays)

        def conv_x2(m, x):
            if (self.gemm != "fp16x2" or not f32 or m.kernel_size != (3, 3) or m.stride != (1, 1) or m.padding != (1, 1) or m.dilation != (1, 1) or m.groups != 1
                    or x.dtype != torch.float32):
                return None
            key = id(m)
            if key not in conv_img:
                conv_img[key] = R.conv3x3_x2_pack(m.weight)
            if conv_img[key] is None:
                return None
            # one workgroup per 16 x 32 output tile: below one tile per CU the library's split-K kernels win (19 x 33 maps with 768 input channels: measured 0.43 vs 0.17 ms)
            if x.shape[0] * ((x.shape[2] + 15) // 16) * ((x.shape[3] + 31) // 32) < 256:
                return None
            return R.conv3x3_x2(x.contiguous(memory_format=CL), conv_img[key], m.out_channels)

        def conv_nb(m, x):   # the module's convolution without its bias
            y = conv_x2(m, x)
            if y is None:
                y = F.conv2d(x, m.weight, None, m.stride, m.padding, m.dilation, m.groups).contiguous(memory_format=CL)
            if self._flop_count is not None:   # flops_per_frame: these calls bypass the modules' forward hooks
                self._flop_count[0] += 2.0 * y.numel() / y.shape[0] * (m.in_channels // m.groups) * m.kernel_size[0] * m.kernel_size[1]
            return y

        def res_unit(unit, x, x_relu=None, extra=None, want_relu=False):
            """PreActResidualLayer: conv2(relu(conv1(relu(x)))) + x [then extra + that]; optionally also relu(result)."""
            x = x.contiguous(memory_format=CL)
            h = torch.relu(x) if x_relu is None else x_relu
            y = R.bias_act(conv_nb(unit.convolution1, h), unit.convolution1.bias, relu=True)
            y = conv_nb(unit.convolution2, y)
            return R.bias_act(y, unit.convolution2.bias, r1=x, r2=extra, want_relu_copy=want_relu)

        for layer in self.model.neck.fusion_stage.layers:
            def fusion_fwd(hidden_state, residual=None, size=None, layer=layer):
                fused = f32 and hidden_state.dtype == torch.float32 and hidden_state.shape[1] % 4 == 0
                if residual is not None:
                    if hidden_state.shape != residual.shape:
                        residual = F.interpolate(residual, size=(hidden_state.shape[2], hidden_state.shape[3]), mode="bilinear",
                                                 align_corners=False)
                    if fused:
                        hidden_state, hr = res_unit(layer.residual_layer1, residual, extra=hidden_state.contiguous(memory_format=CL), want_relu=True)
                        hidden_state = res_unit(layer.residual_layer2, hidden_state, x_relu=hr)
                    else:
                        hidden_state = layer.residual_layer2(hidden_state + layer.residual_layer1(residual))
                else:
                    hidden_state = res_unit(layer.residual_layer2, hidden_state) if fused else layer.residual_layer2(hidden_state)
                tgt = (2 * hidden_state.shape[2], 2 * hidden_state.shape[3]) if size is None else tuple(size)
                if fused and layer.projection.bias is not None:
                    return R.upsample_bilinear_bias(conv_nb(layer.projection, hidden_state), tgt, layer.projection.bias)
                return layer.projection(up(hidden_state, tgt))
            layer.forward = fusion_fwd
        head = self.model.head
        tail = None
        if (f32 and isinstance(head.activation2, torch.nn.ReLU) and head.conv2.out_channels in (16, 32, 64) and head.conv3.kernel_size == (1, 1)
                and head.conv2.bias is not None and head.conv3.bias is not None and head.conv1.bias is not None):
            # conv3's scalar bias is read on the host ONCE per parameter version (no per-call sync); an in-place update of the parameters
            # (load_state_dict, .copy_) bumps their version counters and is picked up at the next call
            tail = {"key": None, "w3": None, "b3": 0.0}

            def tail_operands():
                w, b = head.conv3.weight, head.conv3.bias
                try:
                    key = (w.data_ptr(), w._version, b.data_ptr(), b._version)
                except RuntimeError:   # inference tensors (created / loaded under torch.inference_mode()) track no version counter: immutable outside
                    key = (w.data_ptr(), b.data_ptr())   # inference mode, so the storage address identifies the value
                if tail["key"] != key:
                    tail["w3"] = w.detach().reshape(-1).contiguous()
                    tail["b3"] = float(b.detach().float().item())
                    tail["key"] = key
                return tail["w3"], tail["b3"], float(head.max_depth)

        def head_fwd(hidden_states, patch_height, patch_width):
            x = hidden_states[head.head_in_index]
            size = (int(patch_height * head.patch_size), int(patch_width * head.patch_size))
            if tail is not None and x.dtype == torch.float32:
                h = R.upsample_bilinear_bias(conv_nb(head.conv1, x.contiguous(memory_format=CL)), size, head.conv1.bias)
                y = conv_nb(head.conv2, h)
                if self._flop_count is not None:
                    self._flop_count[0] += 2.0 * y.numel() / y.shape[0]   # the 1x1 convolution to one channel inside the tail kernel
                w3, b3, scale = tail_operands()
                return R.dpt_head_tail(y, head.conv2.bias, w3, b3, scale)
            h = up(head.conv1(x), size)
            h = head.conv3(head.activation1(head.conv2(h)))
            return (head.activation2(h) * head.max_depth).squeeze(dim=1)
        head.forward = head_fwd
        stage = self.model.neck.reassemble_stage

        def reassemble_fwd(hidden_states, patch_height=None, patch_width=None):
            out = []
            for i, hs in enumerate(hidden_states):
                B, _, Cn = hs.shape
                x = hs[:, 1:].reshape(B, patch_height, patch_width, Cn).permute(0, 3, 1, 2)   # NHWC storage, NCHW view: channels_last
                out.append(stage.layers[i](x))
            return out
        stage.forward = reassemble_fwd

    @torch.no_grad()
    def _fuse_backbone_layers(self):
        """Inference-only rewrite of every Dinov2Layer (same math, fewer launches): q/k/v projections as ONE GEMM
        (N = 3*hidden), LayerScale folded into the output-projection / fc2 weights (lambda * (xW^T + b) == x(lambda*W)^T +
        lambda*b), SDPA called directly.  Per layer: 4 GEMMs + attention + 2 LN + GELU + 2 adds instead of 6 GEMMs + 13
        smaller kernels."""
        layers = list(self.model.backbone.encoder.layer)
        R = self.renderer
        # Query-length padding: AOTriton's flash kernel takes a ~1.5x slower path when the QUERY length is not a multiple of
        # 256 (measured on MI355X: Tq=2443 446 us, Tq=2560 302 us per layer, keys unpadded).  DINOv2 always has an odd token
        # count (patches + CLS), so the token sequence is padded ONCE after the embeddings with dummy tokens that are never
        # used as keys/values (k, v are sliced to the real length) and are dropped before the final layer norm: exact.
        pad_state = {"T": None}
        bb = self.model.backbone
        if self.device.type == "cuda":
            def pad_tokens(_mod, _inp, out):
                T = out.shape[1]
                Tp = -(-T // 256) * 256
                pad_state["T"] = None
                if out.dtype == torch.bfloat16 and Tp != T and Tp <= T * 1.1:
                    pad_state["T"] = T
                    return F.pad(out, (0, 0, 0, Tp - T))
                return out
            bb.embeddings.register_forward_hook(pad_tokens)
            final_ln = bb.layernorm.forward
            bb.layernorm.forward = lambda x: final_ln(x if pad_state["T"] is None else x[:, :pad_state["T"]])
        stash = {"x": None, "h": None}   # LayerNorm1(x) computed by the PREVIOUS layer's fused add+LayerNorm launch
        for li, layer in enumerate(layers):
            att, out = layer.attention.attention, layer.attention.output.dense
            wqkv = torch.cat([att.query.weight, att.key.weight, att.value.weight], 0).contiguous()
            bqkv = torch.cat([att.query.bias, att.key.bias, att.value.bias], 0).contiguous()
            l1, l2 = layer.layer_scale1.lambda1.float(), layer.layer_scale2.lambda1.float()
            wo = (out.weight.float() * l1[:, None]).to(out.weight.dtype).contiguous()
            bo = (out.bias.float() * l1).to(out.bias.dtype).contiguous()
            fc1, fc2 = layer.mlp.fc1, layer.mlp.fc2
            w2 = (fc2.weight.float() * l2[:, None]).to(fc2.weight.dtype).contiguous()
            b2 = (fc2.bias.float() * l2).to(fc2.bias.dtype).contiguous()
            nh, hd, scaling = att.num_attention_heads, att.attention_head_size, att.scaling
            n1, n2, act = layer.norm1, layer.norm2, layer.mlp.activation
            nxt = layers[li + 1].norm1 if li + 1 < len(layers) else None
            x3 = None
            if self.gemm != "f32":   # weights split + packed once; exact GELU only (what DINOv2's MLP uses) goes into fc1's epilogue
                gelu_ok = isinstance(act, torch.nn.GELU) and getattr(act, "approximate", "none") == "none" or type(act).__name__ == "GELUActivation"
                gm = self.gemm
                x3 = dict(qkv=(R.gemm_x3_pack(wqkv, gm), wqkv.shape[0]), wo=(R.gemm_x3_pack(wo, gm), wo.shape[0]),
                          fc1=(R.gemm_x3_pack(fc1.weight, gm), fc1.weight.shape[0]), w2=(R.gemm_x3_pack(w2, gm), w2.shape[0]), gelu=bool(gelu_ok), mode=gm)

            def fwd(x, wqkv=wqkv, bqkv=bqkv, wo=wo, bo=bo, fc1=fc1, w2=w2, b2=b2, nh=nh, hd=hd, scaling=scaling, n1=n1, n2=n2,
                    act=act, nxt=nxt, x3=x3):
                B, T, d = x.shape
                if x3 is not None and not (x.dtype == torch.float32 and x.is_contiguous() and d in (384, 768, 1024)):
                    raise NotImplementedError(f"gemm={x3['mode']!r}: hidden size {d} / dtype {x.dtype} not built (float32, 384 / 768 / 1024) -- refusing to fall back silently")
                if x3 is not None:
                    lin = lambda t, key, bias, gelu=False: R.linear_x3(t if t.is_contiguous() else t.contiguous(), x3[key][0], x3[key][1], bias, gelu=gelu,
                                                                       mode=x3["mode"])
                    h = stash["h"] if stash["x"] is x else R.add_layernorm(x, None, n1)[1]
                    stash["x"] = stash["h"] = None
                    qkv = lin(h, "qkv", bqkv)
                    if hd == 64 and not pad_state["T"]:   # the library's split-bf16 attention (every DINOv2 size has 64-wide heads)
                        o = R.attention_x3(qkv, nh, scaling, mode=x3["mode"])
                    else:
                        qkv = qkv.view(B, T, 3, nh, hd)
                        Tk = pad_state["T"] or T
                        q, k, v = qkv[:, :, 0].transpose(1, 2), qkv[:, :Tk, 1].transpose(1, 2), qkv[:, :Tk, 2].transpose(1, 2)
                        o = F.scaled_dot_product_attention(q, k, v, scale=scaling).transpose(1, 2).reshape(B, T, d)
                    a = lin(o, "wo", bo)
                    x, h = R.add_layernorm(x, a, n2)
                    hid = lin(h, "fc1", fc1.bias, gelu=True) if x3["gelu"] else act(lin(h, "fc1", fc1.bias))
                    m = lin(hid, "w2", b2)
                    if nxt is None:
                        return x + m
                    x, hn = R.add_layernorm(x, m, nxt)
                    stash["x"], stash["h"] = x, hn
                    return x
                hip = R is not None and x.dtype in (torch.bfloat16, torch.float32) and x.is_contiguous() and d in (384, 768, 1024)
                if hip:   # residual add + LayerNorm pairs as single HIP launches (vd3d_add_layernorm)
                    h = stash["h"] if stash["x"] is x else R.add_layernorm(x, None, n1)[1]
                    stash["x"] = stash["h"] = None
                else:
                    h = n1(x)
                qkv = F.linear(h, wqkv, bqkv).view(B, T, 3, nh, hd)
                Tk = pad_state["T"] or T   # real tokens only on the key/value side
                q, k, v = qkv[:, :, 0].transpose(1, 2), qkv[:, :Tk, 1].transpose(1, 2), qkv[:, :Tk, 2].transpose(1, 2)
                o = F.scaled_dot_product_attention(q, k, v, scale=scaling).transpose(1, 2).reshape(B, T, d)
                a = F.linear(o, wo, bo)
                if not hip:
                    x = x + a
                    return x + F.linear(act(fc1(n2(x))), w2, b2)
                x, h = R.add_layernorm(x, a, n2)
                m = F.linear(act(fc1(h)), w2, b2)
                if nxt is None:
                    return x + m
                x, hn = R.add_layernorm(x, m, nxt)
                stash["x"], stash["h"] = x, hn
                return x

            layer.forward = fwd

    @torch.no_grad()
    def infer_bgr_u8(self, frames_bgr: torch.Tensor, inference_size=None, raw: bool = False, at_inference_size: bool = False) -> torch.Tensor:
        """uint8 [B,H,W,3] BGR frames in HBM -> float32 [B,H,W] predicted depth at the frame size (the
        depth-estimation pipeline's post-process: bicubic, align_corners=False).  ``raw=True`` returns the model-resolution
        prediction [B,th,tw] instead, for the fused HIP hand-off (Renderer.depth_handoff).  ``at_inference_size``: with an
        ``inference_size`` (W', H') the prediction is returned at (H', W') -- what the transformers pipeline hands back for the
        pre-resized images of hf_batch_safe_pipe (core/render_depth.py:1113-1116)."""
        B, H, W, _ = frames_bgr.shape
        if at_inference_size and inference_size is not None:
            oH, oW = int(inference_size[1]), int(inference_size[0])
        else:
            oH, oW = H, W
        if self.renderer is not None and inference_size is None and frames_bgr.dtype == torch.uint8:
            th, tw = self.resize_target(H, W)
            x = None
            try:
                x = self.renderer.depth_preprocess(frames_bgr, th, tw, self.proc["mean"], self.proc["std"], dtype=self.dtype)
            except Exception as e:   # down-scale beyond the kernel's tap budget: the ATen path below is the same math
                if getattr(e, "code", None) != -4:
                    raise
            if x is not None:
                pred = self.model(pixel_values=x).predicted_depth
                if raw:
                    return pred.float()
                return F.interpolate(pred.float().unsqueeze(1), size=(H, W), mode="bicubic", align_corners=False).squeeze(1)
        x = frames_bgr.to(self.device).flip(-1).permute(0, 3, 1, 2).float()  # RGB, NCHW
        if inference_size is not None:  # hf_batch_safe_pipe: img.resize(inference_size, BICUBIC) first (:1113-1116)
            x = F.interpolate(x, size=(int(inference_size[1]), int(inference_size[0])), mode="bicubic", antialias=True,
                              align_corners=False).clamp_(0, 255)
        th, tw = self.resize_target(x.shape[2], x.shape[3])
        x = F.interpolate(x, size=(th, tw), mode="bicubic", antialias=True, align_corners=False)
        x = ((x / 255.0) - self.mean) / self.std
        x = x.to(self.dtype)
        if self.device.type == "cuda":
            x = x.contiguous(memory_format=torch.channels_last)
        pred = self.model(pixel_values=x).predicted_depth  # [B, th, tw]
        if raw:
            return pred.float()
        pred = F.interpolate(pred.float().unsqueeze(1), size=(oH, oW), mode="bicubic", align_corners=False).squeeze(1)
        return pred

    @torch.no_grad()
    def depth_frames_u8(self, frames_bgr: torch.Tensor, inference_size=None, invert: bool = False) -> torch.Tensor:
        """The depth tab's per-frame product (core/render_depth.py:1907-1917) for uint8 [B,H,W,3] BGR frames: the prediction at the size
        the pipeline saw (the frame, or ``inference_size`` = (W', H')) -> ``convert_depth_to_grayscale`` -> optional ``255 -`` ->
        ``cv2.resize(..., (W, H), INTER_CUBIC)``.  Returns uint8 [B,H,W] on the device (what the reference writes to its depth video)."""
        B, H, W, _ = frames_bgr.shape
        u8 = depth_to_u8(self.infer_bgr_u8(frames_bgr, inference_size, at_inference_size=True), invert)
        if tuple(u8.shape[-2:]) == (H, W):
            return u8
        if self.renderer is None:
            raise RuntimeError("depth_frames_u8 with an inference size needs DepthPipe(renderer=...) for the uint8 INTER_CUBIC resize")
        out = torch.empty((B, H, W), dtype=torch.uint8, device=self.device)
        for b in range(B):
            self.renderer.resize_cubic_u8(u8[b], H, W, out=out[b])
        return out

    def __call__(self, images, inference_size=None):
        """Reference protocol: list of PIL images (or HxWx3 uint8 RGB arrays) -> list of dicts; ``predicted_depth`` has the size
        of the image the pipeline saw (the frame, or ``inference_size`` when given -- the caller then resizes the uint8 map back
        with cv2.INTER_CUBIC, core/render_depth.py:1914-1917)."""
        single = not isinstance(images, (list, tuple))
        imgs = [images] if single else list(images)
        outs = []
        for im in imgs:
            a = np.asarray(im.convert("RGB") if hasattr(im, "convert") else im)
            bgr = torch.from_numpy(np.ascontiguousarray(a[..., ::-1]))[None]
            outs.append({"predicted_depth": self.infer_bgr_u8(bgr, inference_size, at_inference_size=True)[0]})
        return outs

    def flops_per_frame(self, h: int, w: int) -> float:
        """Dense-GEMM flop estimate for MFMA accounting: 2*params_linear*tokens + 4*T^2*d per layer
        for the backbone, the EXECUTED convolution flops of the DPT neck / head counted during one forward."""
        th, tw = self.resize_target(h, w)
        bcfg = getattr(self.model.config, "backbone_config", None) or self.model.config
        ps = int(bcfg.patch_size)
        T = (th // ps) * (tw // ps) + 1
        d, L = int(bcfg.hidden_size), int(bcfg.num_hidden_layers)
        lin = L * (4 * d * d + 8 * d * d)  # qkv+proj, mlp 4x
        backbone = 2.0 * lin * T + 4.0 * T * T * d * L
        total = [0.0]

        def hook(mod, inp, out):
            if isinstance(mod, (torch.nn.Conv2d, torch.nn.ConvTranspose2d)):
                k = mod.kernel_size[0] * mod.kernel_size[1]
                total[0] += 2.0 * out.numel() / out.shape[0] * (mod.in_channels // mod.groups) * k

        hs = [m.register_forward_hook(hook) for m in self.model.modules() if isinstance(m, (torch.nn.Conv2d, torch.nn.ConvTranspose2d))]
        self._flop_count = total   # the bias-free convolution calls of the rewritten neck / head count themselves (EXECUTED flops: the
        try:                       # fusion layers' projections run before their up-samplings, at a quarter of the pixels)
            with torch.no_grad():
                x = torch.zeros(1, 3, th, tw, device=self.device, dtype=self.dtype)
                if self.device.type == "cuda":
                    x = x.contiguous(memory_format=torch.channels_last)
                self.model(pixel_values=x)
        finally:
            self._flop_count = None
            for hnd in hs:
                hnd.remove()
        return backbone + total[0]
