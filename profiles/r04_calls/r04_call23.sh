# round-4 GPU call 23: configs[4] chain on two streams (up-scale net of batch i behind depth + DIBR of batch i + 1) against the serial chain
export TMPDIR=/tmp
O=gpurun_out/c23; mkdir -p $O
timeout 300 python bench.py --upscale-only --chain-serial 2>$O/serial.err | tail -1 > $O/serial.json
timeout 300 python bench.py --upscale-only 2>$O/two.err | tail -1 > $O/two.json
python - <<'PY'
import json
for n in ("serial", "two"):
    try:
        d = json.load(open(f"gpurun_out/c23/{n}.json"))
        print(n, d.get("value"), d.get("ms_per_step"), d.get("stage_ms_per_frame"), d.get("streams"), d.get("error"))
    except Exception as e:
        print(n, "FAILED", e); print(open(f"gpurun_out/c23/{n}.err").read()[-1500:])
PY
