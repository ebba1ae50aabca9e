#!/usr/bin/env python3
"""GPU probe (development aid): what library-side selection is worth on the depth leg of the headline (depth-anything-v2-base, float32,
16 frames of 3840x2160 per batch).
  mode `gemm`  : PyTorch TunableOp over the backbone's four GEMM shapes (hipBLASLt / rocBLAS solution search); the result table is written to
                 argv[2] and the forward is timed before / after; the prediction of the tuned run is compared with the untuned one.
  mode `find`  : MIOpen find mode (torch.backends.cudnn.benchmark) over the DPT neck / head convolutions; first-forward cost and the steady time.
  mode `use`   : forward time with the table argv[2] loaded and tuning off (what DepthPipe does with the committed table).
usage: [MODEL=depth-anything-v2-small] [FRAME=1080x1920] [FIND=1] probe_net_tune.py gemm|find|use [table.csv] [frames per batch]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from visiondepth3d_amd.depth import DepthPipe
from visiondepth3d_amd.render_3d import Renderer

mode = sys.argv[1]
table = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/tunable_gemm.csv"
B = int(sys.argv[3]) if len(sys.argv) > 3 else 16
R = Renderer(0)
if mode == "find" or os.environ.get("FIND") == "1":   # FIND=1: find mode under any mode (e.g. FIND=1 ... use table.csv)
    torch.backends.cudnn.benchmark = True
os.environ["VD3D_TUNED_GEMM"] = "0"      # the probe loads / writes its own table
MODEL = os.environ.get("MODEL", "depth-anything-v2-base")
FH, FW = (int(v) for v in os.environ.get("FRAME", "2160x3840").split("x"))
pipe = DepthPipe(MODEL, device="cuda", dtype=torch.float32, renderer=R)
x = torch.randint(0, 255, (B, FH, FW, 3), dtype=torch.uint8, device="cuda", generator=torch.Generator("cuda").manual_seed(1))


def timed(iters=4):
    for _ in range(2):
        pipe.infer_bgr_u8(x, raw=True)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(iters):
        p = pipe.infer_bgr_u8(x, raw=True)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / iters * 1e3, p


t0 = time.perf_counter()
with torch.no_grad():
    pipe.infer_bgr_u8(x, raw=True)
torch.cuda.synchronize()
print(f"[{mode}] first forward {time.perf_counter() - t0:.1f} s", flush=True)
with torch.no_grad():
    ms0, ref = timed()
print(f"[{mode}] forward {ms0:.2f} ms / {B} frames", flush=True)
if mode in ("gemm", "use"):
    import torch.cuda.tunable as tn
    tn.enable(True)
    tn.set_filename(table)
    if mode == "gemm":
        tn.tuning_enable(True)
        tn.set_max_tuning_duration(int(os.environ.get("TUNE_MS", "40")))
        tn.set_max_tuning_iterations(int(os.environ.get("TUNE_ITERS", "30")))
        t0 = time.perf_counter()
        with torch.no_grad():
            pipe.infer_bgr_u8(x, raw=True)
        torch.cuda.synchronize()
        print(f"[gemm] tuning forward {time.perf_counter() - t0:.1f} s", flush=True)
        tn.tuning_enable(False)
        res = tn.get_results()
        print(json.dumps({"validators": tn.get_validators(), "results": res}, indent=0)[:6000], flush=True)
        with open(table + ".json", "w") as f:
            json.dump({"validators": tn.get_validators(), "results": res}, f, indent=1)
    else:
        tn.tuning_enable(False)
        print("[use] read_file:", tn.read_file(table), flush=True)
    with torch.no_grad():
        ms1, p1 = timed()
    d = (p1 - ref).abs().max().item() / ref.abs().max().item()
    print(f"[{mode}] forward with the table {ms1:.2f} ms ({ms0 / ms1:.3f}x); max |pred - untuned| / max |untuned| = {d:.3e}", flush=True)
