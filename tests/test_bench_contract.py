"""bench.py's record assembly without a GPU: the roofline objects and sub-records built from a synthetic measurement must carry the
fields the driver contract names (roofline: bound / achieved / peak / unit / frac / traffic), stay JSON-serialisable, and price the
kernels with SURVEY 8(d)'s algorithmic bytes."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _res(**over):
    r = dict(workload="4k-dibr", desc="d", sh=2160, sw=3840, model=None, B=16, steps=13, warmup=2, dt=0.18, frames_total=208,
             stage_ms={"w1": 0.16, "finish": 0.47, "p1_own": 0.14, "p3_own": 0.6, "warp": 0.25, "replay": 0.04},
             iso_ms={"w1": 0.144, "finish": 0.42, "frame": 0.99}, net_ms=None, flops_per_frame=None, N=3840 * 2160, pix_ov=True,
             depth_dtype=None, host_io=False)
    r.update(over)
    return r


def test_roofline_objects_and_sub_records():
    import bench
    rf = bench.rooflines(_res(), copy_gbs=4600.0)
    json.dumps(rf)
    w1 = rf["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in w1
    n = 3840 * 2160
    assert w1["algorithmic_bytes_per_launch"] == 13 * n and w1["unit"] == "GB/s" and w1["peak"] == 8000.0
    # the kernel's cost is its duration in the SEQUENTIAL pass (0.144 ms); the in-step duration (0.16 ms, CUs shared with the other streams) stays beside it
    assert abs(w1["achieved"] - 13 * n / 0.144e-3 / 1e9) < 0.1 and abs(w1["frac"] - w1["achieved"] / 8000.0) < 1e-4
    assert w1["avg_launch_ms"] == 0.144 and w1["in_step_avg_launch_ms"] == 0.16 and abs(w1["in_step_frac"] - 13 * n / 0.16e-3 / 1e9 / 8000.0) < 1e-4
    assert w1["bound"] in ("valu", "hbm") and w1["avg_launch_measured"].startswith("sequential")
    e1 = rf["roofline_e1"]
    assert e1["algorithmic_bytes_per_launch"] == 10 * n and abs(e1["frac"] - 10 * n / 0.42e-3 / 1e9 / 8000.0) < 1e-4
    assert abs(e1["in_step_frac"] - 10 * n / 0.47e-3 / 1e9 / 8000.0) < 1e-4
    # a run without the sequential pass prices the in-step duration and says so
    r0 = bench.rooflines(_res(iso_ms={}), copy_gbs=4600.0)["roofline"]
    assert r0["avg_launch_ms"] == 0.16 and r0["avg_launch_measured"].startswith("inside the timed region")
    assert rf["roofline_chain"]["algorithmic_bytes_per_frame"] == 17 * n
    # a workload with a depth net reports the MFMA fraction against the dtype's dense peak
    rd = bench.rooflines(_res(workload="4k-dav2b-dibr", model="depth-anything-v2-base", net_ms=140.0, flops_per_frame=7.9e11, depth_dtype="f32"))
    dn = rd["roofline_depthnet"]
    assert dn["bound"] == "mfma" and dn["peak"] == 157.3 and abs(dn["achieved"] - 7.9e11 * 16 / 0.14 / 1e12) < 0.01
    sr = bench.sub_record(_res())
    assert sr["unit"] == "stereo-pairs/s" and abs(sr["value"] - 208 / 0.18) < 1e-2 and sr["dtype"] == "f32"
    assert "REDUCED" in bench.sub_record(_res(depth_dtype="bf16"))["dtype"]


def test_workload_table_names_the_baseline_configs():
    import bench
    assert bench.HEADLINE == "4k-dav2b-dibr" and bench.HEADLINE in bench.WORKLOADS
    for name in ("1080p-dav2s-dibr", "4k-dibr", "1080p-dibr", "4k-dibr-sepdof"):
        assert name in bench.WORKLOADS
    assert bench.WORKLOADS["4k-dibr"][:2] == (2160, 3840) and bench.WORKLOADS["1080p-dav2s-dibr"][2] == "depth-anything-v2-small"
