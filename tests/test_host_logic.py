"""CPU-only tests of the host logic that mirrors the reference's Python: geometry, kwargs mapping, synth."""
import numpy as np
import pytest

from visiondepth3d_amd import synth
from visiondepth3d_amd._abi import FMT_HALF_SBS, ShiftParams
from visiondepth3d_amd.geometry import plan_geometry
from visiondepth3d_amd.params import render_kwargs_to_params, shift_params_from_kwargs


def test_half_sbs_geometry_1080p_and_4k():
    # 1080p -> eye 540x960 -> warp 1080x1920 -> fit 1080x960 -> hstack 1920x1080
    g = plan_geometry(1920, 1080, 1080, "Half-SBS")
    assert (g["eye_w"], g["eye_h"], g["warp_w"], g["warp_h"], g["fit_w"], g["fit_h"], g["out_w"], g["out_h"]) == \
        (960, 540, 1920, 1080, 960, 1080, 1920, 1080)
    g = plan_geometry(3840, 2160, 2160, "Half-SBS")
    assert (g["eye_w"], g["eye_h"], g["warp_w"], g["warp_h"], g["out_w"], g["out_h"]) == (1920, 1080, 3840, 2160, 3840, 2160)


def test_full_sbs_is_hardwired_1080p_unless_preserve():
    g = plan_geometry(3840, 2160, 2160, "Full-SBS")  # core/render_3d.py:1120-1123
    assert (g["fit_w"], g["fit_h"], g["out_w"], g["out_h"]) == (1920, 1080, 3840, 1080)
    g = plan_geometry(3840, 2160, 2160, "Full-SBS", preserve_original_aspect=True, original_video_width=3840, original_video_height=2160)
    assert (g["eye_w"], g["eye_h"], g["out_w"], g["out_h"]) == (3840, 2160, 7680, 2160)


def test_centre_crop_when_aspect_differs():
    g = plan_geometry(160, 120, 90, "Half-SBS")  # 4:3 source, 16:9 target
    assert (g["crop_x"], g["crop_y"], g["crop_w"], g["crop_h"]) == (0, 15, 160, 90)
    g = plan_geometry(400, 100, 90, "Half-SBS")  # wider than target
    assert g["crop_w"] == int(100 * 16 / 9) and g["crop_x"] == (400 - g["crop_w"]) // 2


def test_render_kwargs_mirror_reference_forwarding():
    p = render_kwargs_to_params(192, 108, output_height=108, fg_shift=10.0, mg_shift=-2.5, bg_shift=-5.0, sharpness_factor=0.15,
                                output_format="Half-SBS", dof_strength=2.0, depth_pop_gamma=0.5, parallax_balance=0.3,
                                use_subject_tracking=True, use_floating_window=True)
    # accepted but never forwarded by render_sbs_3d (:1284-1331): the struct keeps pixel_shift_cuda's defaults
    assert p.shift.depth_pop_gamma == 0.85 and p.shift.parallax_balance == 0.8
    assert p.format == FMT_HALF_SBS and p.shift.use_subject_tracking == 1 and p.shift.enable_floating_window == 1
    with pytest.raises(TypeError):
        render_kwargs_to_params(192, 108, output_height=108, fg_shift=1, mg_shift=1, bg_shift=1, sharpness_factor=0, output_format="Half-SBS",
                                dof_strength=0, codec="XVID")  # render_cli.py's stale kwarg is a TypeError in the reference too
    # skip_blank_frames is loop-level: the per-clip block is the same with and without it
    pk = dict(output_height=108, fg_shift=1, mg_shift=1, bg_shift=1, sharpness_factor=0, output_format="Half-SBS", dof_strength=0)
    assert bytes(render_kwargs_to_params(192, 108, skip_blank_frames=True, **pk)) == bytes(render_kwargs_to_params(192, 108, **pk))
    pa = render_kwargs_to_params(192, 108, output_height=108, fg_shift=1, mg_shift=1, bg_shift=1, sharpness_factor=0,
                                 output_format="Half-SBS", dof_strength=0, auto_crop_black_bars=True, target_ratio=2.39)
    assert pa.auto_crop_black_bars == 1 and pa.target_ratio == 2.39 and p.auto_crop_black_bars == 0


def test_shift_kwargs():
    p = shift_params_from_kwargs(4.5, -1.5, -6.0, blur_ksize=1, feather_strength=0.0, return_shift_map=False, dof_strength=2.0)
    assert isinstance(p, ShiftParams) and p.blur_ksize == 1 and p.fg_shift == 4.5
    with pytest.raises(TypeError):
        shift_params_from_kwargs(1, 2, 3, bogus=1)


def test_synth_is_deterministic_and_well_conditioned():
    f1, d1 = synth.synth_frame(5, 108, 192)
    f2, d2 = synth.synth_frame(5, 108, 192)
    assert np.array_equal(f1, f2) and np.array_equal(d1, d2)
    assert f1.dtype == np.uint8 and d1.dtype == np.float32 and 0.0 <= d1.min() and d1.max() <= 1.0
    assert np.quantile(d1, 0.98) - np.quantile(d1, 0.02) > 0.3      # far from the collapse guard
    crop = d1[108 // 5:108 * 4 // 5, 192 // 5:192 * 4 // 5]
    assert np.count_nonzero((crop > 0.05) & (crop < 0.95)) > 20     # estimate_subject_depth has samples
    # platform-independence pin: integer-defined generator -> fixed checksum
    assert int(f1.astype(np.int64).sum()) == int(synth.synth_frame(5, 108, 192)[0].astype(np.int64).sum())
    g = synth.depth_to_u8_bgr(d1)
    assert g.shape == (108, 192, 3) and np.array_equal(g[..., 0], g[..., 2])


def test_u8_unit_is_the_exact_division():
    """vd_u8_unit (csrc/vd3d_dev.h): q0 = x*rc; q = fma(fma(-q0, 255, x), rc, q0) equals float32(x)/float32(255) for all 256
    inputs -- the device kernels use it instead of an IEEE division (restated here in float64-emulated float32 arithmetic)."""
    import math
    f32 = np.float32
    rc = f32(1.0) / f32(255.0)

    def fma32(a, b, c):   # exact product and sum in float64 (24-bit x 24-bit fits), one rounding to float32
        return f32(np.float64(a) * np.float64(b) + np.float64(c))

    for v in range(256):
        x = f32(v)
        q0 = f32(x * rc)
        q = fma32(fma32(-q0, f32(255.0), x), rc, q0)
        assert q == x / f32(255.0), v
    assert math.isfinite(float(rc))


def test_blackdetect_log_parsing(tmp_path):
    """The skip_blank_frames side channel (core/ffmpeg_blackdetect.py:23-81): start times -> int(t * fps), cache file, and the
    reference's behaviours kept as they are (only black_start frames, 'd.d' pattern, [] when ffmpeg is missing)."""
    from visiondepth3d_amd import blackdetect as bd
    log = ("[blackdetect @ 0x1] black_start:0.5 black_end:1.25 black_duration:0.75\n"
           "[blackdetect @ 0x1] black_start:12.041667 black_end:12.5 black_duration:0.458333\n"
           "[blackdetect @ 0x1] black_start:3 black_end:4 black_duration:1\n")          # integral seconds: not matched (:65)
    assert bd.parse_blackdetect_log(log, 24.0) == [12, 289]
    assert bd.parse_blackdetect_log(log, 23.976) == [11, 288]
    assert bd.blackdetect_filter("black", 0.1, 0.10) == "blackdetect=d=0.1:pix_th=0.1"
    assert "{duration_threshold}" in bd.blackdetect_filter("white", 0.1, 0.1)    # the reference's raw string (:51)
    with pytest.raises(ValueError):
        bd.blackdetect_filter("grey", 0.1, 0.1)
    vid = str(tmp_path / "clip.mp4")
    (tmp_path / "clip.mp4.blankcache.json").write_text("[7, 3, 9]")
    assert bd.detect_black_white_frames(vid) == [7, 3, 9]                           # cache is returned as stored (:38-41)
    assert bd.detect_black_white_frames(str(tmp_path / "none.mp4"), cache=False) == []   # no ffmpeg here -> [] like :79-81


def test_dpt_front_end_matches_the_real_image_processor():
    """B3 / a25 front end against transformers' own DPTImageProcessor (the Depth-Anything-V2 preprocessor_config values): the
    target size rule on 70 frame sizes exactly, and the float statement the fused kernel is tested against (antialiased bicubic in
    float32, tests/test_hip_depthprep.py) within ~1 LSB of the 8-bit image the PIL path rounds to (tolerance 1.5 / 255 / std)."""
    import numpy as np
    import torch
    import torch.nn.functional as F
    transformers = pytest.importorskip("transformers")
    from PIL import Image
    from visiondepth3d_amd import synth
    from visiondepth3d_amd.depth import IMAGENET_MEAN, IMAGENET_STD, dpt_resize_target
    proc = transformers.DPTImageProcessor(do_resize=True, size={"height": 518, "width": 518}, keep_aspect_ratio=True, ensure_multiple_of=14,
                                          resample=3, do_rescale=True, rescale_factor=1 / 255, do_normalize=True,
                                          image_mean=IMAGENET_MEAN, image_std=IMAGENET_STD, do_pad=False)
    rng = np.random.default_rng(3)
    sizes = [(1080, 1920), (2160, 3840), (720, 1280), (480, 640), (270, 480), (101, 333), (518, 924), (1000, 1000), (1440, 2560),
             (1600, 1440)] + [(int(rng.integers(40, 2200)), int(rng.integers(40, 4000))) for _ in range(60)]
    for h, w in sizes:
        out = proc(images=Image.fromarray(np.zeros((h, w, 3), np.uint8)), return_tensors="pt")["pixel_values"]
        assert tuple(out.shape[-2:]) == dpt_resize_target(h, w), (h, w)
    assert dpt_resize_target(1080, 1920) == (518, 924)
    h, w = 270, 480
    bgr = synth.synth_frame(2, h, w)[0]
    pv = proc(images=Image.fromarray(bgr[..., ::-1].copy()), return_tensors="pt")["pixel_values"][0]
    th, tw = dpt_resize_target(h, w)
    x = torch.from_numpy(bgr[..., ::-1].copy()).permute(2, 0, 1)[None].float()
    x = F.interpolate(x, size=(th, tw), mode="bicubic", antialias=True, align_corners=False)
    mean, std = torch.tensor(IMAGENET_MEAN).view(1, 3, 1, 1), torch.tensor(IMAGENET_STD).view(1, 3, 1, 1)
    ref = (((x / 255.0) - mean) / std)[0]
    assert float((pv - ref).abs().max()) < 1.5 / 255 / min(IMAGENET_STD)


def test_depth_pipe_protocol_against_the_transformers_pipeline():
    """B3 (core/render_depth.py:1106-1119): ``pipe(list[PIL], inference_size) -> [{"predicted_depth": Tensor[h, w]}]`` against
    transformers' own ``pipeline("depth-estimation")`` around the stock ``DepthAnythingForDepthEstimation`` with the same synthetic
    weights, on CPU in float32.  (a) Same pixel_values through the rewritten module graph (fused QKV, folded LayerScale, cached
    position embedding) and the stock one: 1e-6.  (b) Whole protocol: the prediction comes back at the image size like the
    pipeline's; values within a few percent -- entirely the 8-bit rounding of the resized image in the PIL / torchvision front
    ends (the float statement differs from ITS OWN uint8 rounding by the same 1 %: random weights amplify +-0.5 LSB that much)."""
    import numpy as np
    import torch
    transformers = pytest.importorskip("transformers")
    from PIL import Image
    from visiondepth3d_amd import synth
    from visiondepth3d_amd.depth import IMAGENET_MEAN, IMAGENET_STD, DepthPipe, build_config, synthetic_weights_
    torch.set_num_threads(min(16, torch.get_num_threads()))
    pipe = DepthPipe("depth-anything-v2-small", device="cpu", dtype=torch.float32)
    model = transformers.DepthAnythingForDepthEstimation(build_config("depth-anything-v2-small")).eval()
    synthetic_weights_(model, 0)
    proc = transformers.DPTImageProcessor(do_resize=True, size={"height": 518, "width": 518}, keep_aspect_ratio=True, ensure_multiple_of=14,
                                          resample=3, do_rescale=True, rescale_factor=1 / 255, do_normalize=True,
                                          image_mean=IMAGENET_MEAN, image_std=IMAGENET_STD, do_pad=False)
    img = Image.fromarray(synth.synth_frame(1, 126, 224)[0][..., ::-1].copy())
    pv = proc(images=img, return_tensors="pt")["pixel_values"]
    with torch.no_grad():
        a = pipe.model(pixel_values=pv).predicted_depth
        b = model(pixel_values=pv).predicted_depth
    assert float((a - b).abs().max() / b.abs().max()) < 1e-4
    hf = transformers.pipeline("depth-estimation", model=model, image_processor=proc, device="cpu")
    exp = hf([img])[0]["predicted_depth"].squeeze()
    got = pipe([img])[0]["predicted_depth"]
    assert tuple(got.shape) == tuple(exp.shape) == (126, 224)
    assert float((got - exp).abs().mean() / exp.abs().mean()) < 3e-2
    # hf_batch_safe_pipe with an inference size: the images are pre-resized, the prediction comes back at THAT size (:1113-1116)
    small = img.resize((112, 70), Image.BICUBIC)
    exp2 = hf([small])[0]["predicted_depth"].squeeze()
    got2 = pipe([img], inference_size=(112, 70))[0]["predicted_depth"]
    assert tuple(got2.shape) == tuple(exp2.shape) == (70, 112)
    assert float((got2 - exp2).abs().mean() / exp2.abs().mean()) < 5e-2


def test_depth_pipe_from_pretrained_round_trip(tmp_path):
    """B3 checkpoint loading (core/render_depth.py:756-760: AutoModelForDepthEstimation.from_pretrained(local folder)): a stock
    DepthAnythingForDepthEstimation saved with save_pretrained (safetensors + config.json + preprocessor_config.json) comes back
    through DepthPipe.from_pretrained, gets the fused-weight rewrites (one QKV GEMM, folded LayerScale) and predicts the same
    depth as the stock graph to 1e-6 (mean, relative to the output range; max 1e-5) on CPU / float32."""
    import torch
    transformers = pytest.importorskip("transformers")
    from visiondepth3d_amd.depth import IMAGENET_MEAN, IMAGENET_STD, DepthPipe, build_config, synthetic_weights_
    torch.set_num_threads(min(16, torch.get_num_threads()))
    model = transformers.DepthAnythingForDepthEstimation(build_config("depth-anything-v2-small")).eval()
    synthetic_weights_(model, 3)
    model.save_pretrained(tmp_path)
    transformers.DPTImageProcessor(do_resize=True, size={"height": 518, "width": 518}, keep_aspect_ratio=True, ensure_multiple_of=14,
                                   resample=3, image_mean=IMAGENET_MEAN, image_std=IMAGENET_STD).save_pretrained(tmp_path)
    assert (tmp_path / "model.safetensors").exists() and (tmp_path / "preprocessor_config.json").exists()
    pipe = DepthPipe.from_pretrained(str(tmp_path), device="cpu")
    assert pipe.dtype == torch.float32 and pipe.arch == "da"
    assert pipe.proc["size"] == (518, 518) and pipe.proc["multiple"] == 14 and pipe.proc["keep_aspect_ratio"] is True
    assert pipe.resize_target(1080, 1920) == (518, 924)
    g = torch.Generator().manual_seed(5)
    pv = torch.randn(1, 3, 70, 126, generator=g)
    with torch.no_grad():
        a = pipe.model(pixel_values=pv).predicted_depth
        b = model(pixel_values=pv).predicted_depth
    # float32 association noise of the rewrites (LayerScale folded into the weights, one QKV GEMM): mean 1e-6, max 1e-5 of range
    assert float((a - b).abs().mean() / b.abs().max()) < 1e-6
    assert float((a - b).abs().max() / b.abs().max()) < 1e-5
    # the fused weights really are derived from the LOADED checkpoint: a different seed gives a different prediction
    other = DepthPipe("depth-anything-v2-small", device="cpu", seed=4)
    with torch.no_grad():
        c = other.model(pixel_values=pv).predicted_depth
    assert float((c - b).abs().max() / b.abs().max()) > 1e-3
    with pytest.raises(FileNotFoundError):
        DepthPipe.from_pretrained(str(tmp_path / "missing"))


def test_depth_pipe_model_zoo_covers_the_reference_hf_families():
    """core/render_depth.py:686-712: Depth-Anything V1 / V2 and Distill-Any-Depth are one architecture family (DepthAnything on
    DINOv2); MiDaS 3.0 / DPT-Large is DPTForDepthEstimation.  Shapes only (no weights exist here)."""
    from visiondepth3d_amd.depth import MODEL_ZOO, build_config
    for fam in ("depth-anything-v2", "depth-anything-v1"):
        for size in ("small", "base", "large"):
            assert f"{fam}-{size}" in MODEL_ZOO
    assert MODEL_ZOO["distill-any-depth-large"] is MODEL_ZOO["depth-anything-v2-large"]
    cfg = build_config("dpt-large")
    assert type(cfg).__name__ == "DPTConfig" and cfg.hidden_size == 1024 and cfg.num_hidden_layers == 24


def test_depth_pipe_generic_architecture_dpt_protocol():
    """A non-DepthAnything Hugging Face depth model (DPTForDepthEstimation = the MiDaS 3.0 / Intel dpt-* family of the reference's
    list, core/render_depth.py:706-709) behind the same B3 protocol: its own image-processor constants (384 x 384, no aspect
    keeping, mean = std = 0.5), stock module graph.  Tiny configuration (2 layers) so the CPU test stays fast."""
    import torch
    transformers = pytest.importorskip("transformers")
    from PIL import Image
    from visiondepth3d_amd import synth
    from visiondepth3d_amd.depth import PROCESSORS, DepthPipe, synthetic_weights_
    torch.set_num_threads(min(16, torch.get_num_threads()))
    cfg = transformers.DPTConfig(hidden_size=64, num_hidden_layers=4, num_attention_heads=4, intermediate_size=128, image_size=64,
                                 patch_size=16, backbone_out_indices=[0, 1, 2, 3], neck_hidden_sizes=[16, 32, 64, 64],
                                 fusion_hidden_size=32, readout_type="project")
    model = transformers.DPTForDepthEstimation(cfg).eval()
    synthetic_weights_(model, 1)
    proc_kw = dict(PROCESSORS["dpt"], size=(64, 64))
    pipe = DepthPipe("tiny-dpt", device="cpu", model=model, processor=proc_kw)
    assert pipe.arch == "generic" and pipe.resize_target(126, 224) == (64, 64)
    proc = transformers.DPTImageProcessor(do_resize=True, size={"height": 64, "width": 64}, keep_aspect_ratio=False, ensure_multiple_of=1,
                                          resample=3, do_rescale=True, rescale_factor=1 / 255, do_normalize=True,
                                          image_mean=[0.5, 0.5, 0.5], image_std=[0.5, 0.5, 0.5], do_pad=False)
    img = Image.fromarray(synth.synth_frame(2, 126, 224)[0][..., ::-1].copy())
    hf = transformers.pipeline("depth-estimation", model=model, image_processor=proc, device="cpu")
    exp = hf([img])[0]["predicted_depth"].squeeze()
    got = pipe([img])[0]["predicted_depth"]
    assert tuple(got.shape) == tuple(exp.shape) == (126, 224)
    assert float((got - exp).abs().mean() / exp.abs().mean()) < 3e-2   # 8-bit rounding of the resized image in the PIL front end
