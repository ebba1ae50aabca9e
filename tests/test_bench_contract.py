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
    assert r0["avg_launch_ms"] == 0.16 and r0["avg_launch_measured"].startswith("profiled pass behind the timed region")
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


def _synthetic_full_record():
    """A full record assembled without a GPU from bench.py's own builders (rooflines / sub_record), with every sub-record the default run
    produces, each carrying the heaviest extras (two roofline objects + notes) -- larger than any real run's."""
    import bench
    rf = bench.rooflines(_res(), copy_gbs=6200.0)
    rd = bench.rooflines(_res(workload="4k-dav2b-dibr", model="depth-anything-v2-base", net_ms=119.0, flops_per_frame=7.83e11, depth_dtype="f32"))
    res = {"metric": "stereo-pairs/sec end-to-end (depth+warp+fill+mux)", "value": 133.792, "unit": "stereo-pairs/s", "n_gpus": 8, "steps": 20, "warmup": 5,
           "ms_per_step": 119.589, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic " + "x" * 300,
           "config": {"workload": "4k-dav2b-dibr", "description": "d" * 400, "frame": "3840x2160", "format": "Half-SBS", "frames_per_step": 16,
                      "depth_model": "depth-anything-v2-base", "depth_net_dtype": "float32 (the reference's precision)", "pixel_overlap": True, "pixel_streams": 2,
                      "rccl_ranks": 8, "rank_pids": [1000000 + i for i in range(8)], "p1_chain_wait_ms_per_step_min_max": [0.123456, 9.876543],
                      "comm_per_step_per_rank": 8294400 + 640, "clip_frames_all_ranks": 256, "params": "p" * 300},
           "stage_ms": {k: 1.0 for k in "abcdefghij"}, "commit": "abcdef0"}
    res.update(rf)
    res["roofline_depthnet"] = rd["roofline_depthnet"]
    for k in ("roofline", "roofline_e1"):
        res[k]["traffic"] = 362719869
        res[k]["valu"] = {"lane_instr_per_pixel": 553.8, "measured_v_fma_rate_lane_ops_per_s": 5.02e13, "spec_lane_ops_per_s": 7.86e13,
                          "frac_of_measured_rate": 0.59, "frac_of_spec_rate": 0.38, "source": "profiles/x.md"}
        res[k]["measured_on"] = "sub-record 4k-dibr (208 timed frames, DIBR only)"
    names = ["4k-dibr", "1080p-dav2s-dibr", "1080p-dibr", "4k-dav2b-dibr-bf16", "4k-dav2b-dibr-f32x3", "4k-dibr-sepdof", "4k-dibr-dof3", "4k-dibr-anaglyph",
             "4k-dibr-hostio", "1080p-gui-defaults", "4k-dibr-gui", "4k-dibr-vr", "4k-dibr-hostio-nv12", "1080p-dav2s-dibr-esrgan4k", "extra-1", "extra-2"]
    res["sub_records"] = {n: bench.sub_record(_res(workload=n, desc="y" * 400, depth_dtype="bf16"),
                                              {"roofline": dict(rf["roofline"]), "roofline_e1": dict(rf["roofline_e1"]), "note": "n" * 1500,
                                               "stage_ms": {k: 1.0 for k in "abcdefghij"}}) for n in names}
    cb = {"value": 2.21, "unit": "stereo-pairs/s", "cores": 128, "host_threads": 128, "kind": "port", "sample": "s" * 600,
          "single_core": {"value": 0.1665, "unit": "stereo-pairs/s", "cores": 1, "kind": "port", "sample": "s" * 300},
          "gpu_same_work": {"workload": "4k-dibr", "value": 1915.98, "unit": "stereo-pairs/s", "note": "n" * 200}}
    res["cpu_baseline"] = cb
    res["cpu_baseline_1080p"] = dict(cb)
    res["cpu_depth_net"] = {"value": 0.3, "note": "n" * 200}
    return res


def test_compact_line_is_small_and_carries_the_contract():
    """Round 5's 22 KB line came back `parsed: null` from the driver (it reads an 8 KB tail): the final line stays far below that however
    large the full record grows, and still carries every field the contract names."""
    import bench
    res = _synthetic_full_record()
    assert len(json.dumps(res)) > 40000   # the full record IS large
    c = bench.compact_record(res, "gpurun_out/bench_full.json")
    line = json.dumps(c, separators=(",", ":"))
    assert len(line) < bench.COMPACT_LINE_LIMIT < 8192, len(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline"):
        assert k in c, k
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in c["roofline"], k
    assert abs(c["roofline"]["frac"] - c["roofline"]["achieved"] / c["roofline"]["peak"]) < 1e-4
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c["cpu_baseline"], k
    assert c["config"]["workload"] == "4k-dav2b-dibr" and "model" not in c["config"]
    # the multi-rank fields stay in the compact line, so that the first SCALE run explains itself
    assert c["config"]["rccl_ranks"] == 8 and len(c["config"]["rank_pids"]) == 8 and c["config"]["p1_chain_wait_ms_per_step_min_max"][1] > 9
    assert set(c["sub_records"]) == set(res["sub_records"])
    for r in c["sub_records"].values():
        assert set(r) <= {"value", "ms_per_step", "dtype", "w1_frac", "w1_avg_launch_ms"} and "value" in r
    assert c["roofline_e1"]["frac"] == res["roofline_e1"]["frac"] and c["roofline_depthnet"]["bound"] == "mfma"
    # a real full record of an earlier round compacts too
    old = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r05_bench_default.json")))
    assert len(json.dumps(bench.compact_record(old), separators=(",", ":"))) < bench.COMPACT_LINE_LIMIT


def test_emitted_last_stdout_line_parses_and_is_the_compact_one(tmp_path, capsys):
    import bench
    res = _synthetic_full_record()
    bench.emit_record(res, str(tmp_path / "full.json"))
    out = capsys.readouterr().out
    lines = [ln for ln in out.splitlines() if ln.strip()]
    last = json.loads(lines[-1])
    assert len(lines[-1]) < 8192 and last["value"] == res["value"] and "roofline" in last and "cpu_baseline" in last
    # what a tail-reading driver sees: the last 8 KB of stdout still hold the whole line
    assert out[-8192:].rstrip("\n").endswith(lines[-1]) and lines[-1] in out[-8192:]
    # the full record: an earlier line (wrapped) and the file
    assert json.loads(lines[-2])["bench_full_record"]["sub_records"]["4k-dibr"]["note"] == "n" * 1500
    assert json.load(open(tmp_path / "full.json"))["value"] == res["value"]
