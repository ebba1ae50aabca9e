#!/usr/bin/env python3
"""GPU probe: RCCL collectives of the sharded bench path on a single-rank process group -- dtype coverage (uint8 planes, float32 and
int64 records, int32 crop rectangles) and ordering on the renderer's private (external) HIP stream."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
from visiondepth3d_amd.render_3d import Renderer
r = Renderer(0, private_stream=True)
s = r.stream
for dt, shape in ((torch.uint8, (4, 270, 480)), (torch.float32, (4, 2)), (torch.int64, (4, 4)), (torch.int32, (4, 4))):
    with torch.cuda.stream(s):
        src = (torch.arange(int(torch.tensor(shape).prod()), device="cuda") % 251).to(dt).view(shape)
        out = torch.empty_like(src)
        dist.all_gather_into_tensor(out, src)
        ok = bool((out == src).all())
    print(dt, shape, "ok" if ok else "MISMATCH")
dist.barrier()
dist.destroy_process_group()
print("done")
