"""Does a depth-net run in an x3 mode slow down later DIBR-only workloads of the same process?  (round 6 investigation)"""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

args = types.SimpleNamespace(gpus=1, steps=6, warmup=2, batch=16, clip=32, depth_dtype="f32", no_cpu_baseline=True, no_sub_records=True, no_profile=False,
                             no_miopen_find=False, sharded=False, per_frame=False, host_io=False, host_io_nv12=False, ring_depth=4, pixel_overlap=None, pix_streams=2,
                             no_overlap=False, upscale_only=False, chain_serial=False, workload=None)
env = bench.Env(args)
def gui():
    r = bench.run_workload(env, args, "4k-dibr-gui", 6, 2, profile=True)
    return round(r["frames_total"] / r["dt"], 1)
print("gui before:", gui(), gui())
for mode in sys.argv[1:] or ["f32"]:
    r = bench.run_workload(env, args, bench.HEADLINE, 4, 2, depth_dtype=mode, profile=True, isolated_pass=False)
    print("headline", mode, round(r["frames_total"] / r["dt"], 1))
    print("gui after", mode, ":", gui(), gui())
