#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/c7
export PYTHONUNBUFFERED=1
./tools/malloc_probe two > gpurun_out/c7/malloc.txt 2>&1
./tools/malloc_probe one >> gpurun_out/c7/malloc.txt 2>&1
./tools/malloc_probe small >> gpurun_out/c7/malloc.txt 2>&1
cat gpurun_out/c7/malloc.txt
PECANPY_AMD_CREATE_DEBUG=1 python3 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/c7/bench.json 2> gpurun_out/c7/bench.err
grep "create\]" gpurun_out/c7/bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/c7/bench.json").read().strip().splitlines()[-1])
print({k:d[k] for k in ("value","ms_per_step")}, {k:v for k,v in d["config"].items() if "index" in k or "build" in k or "create" in k or "first" in k})
PY
timeout 600 python -m pytest tests/test_gpu_sharding.py tests/test_gpu_lanes.py -m gpu -x -q -k "peer_row or two_ranks" > gpurun_out/c7/t.log 2>&1; echo "t rc=$?"; tail -15 gpurun_out/c7/t.log
