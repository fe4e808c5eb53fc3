#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/c6
export PYTHONUNBUFFERED=1
# exactly the driver's command, first thing on the fresh box
PECANPY_AMD_CREATE_DEBUG=1 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/c6/bench.json 2> gpurun_out/c6/bench.err
grep "create\]" gpurun_out/c6/bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/c6/bench.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step")}, {k:v for k,v in d["config"].items() if "index" in k or "build" in k or "create" in k})
print(d["roofline"].get("frac"), d["cpu_baseline"]["value"])
PY
# again in a second process (warm box)
PECANPY_AMD_CREATE_DEBUG=1 python3 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/c6/bench2.json 2> gpurun_out/c6/bench2.err
grep "create\]" gpurun_out/c6/bench2.err
