#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/c9
export PYTHONUNBUFFERED=1
timeout 400 python -m pytest tests/test_gpu_lane_index.py -m gpu -x -q -s -k "partial" > gpurun_out/c9/partial.log 2>&1; echo "partial rc=$?"; grep "partial\]" gpurun_out/c9/partial.log; tail -4 gpurun_out/c9/partial.log | cut -c1-300
timeout 300 python tools/ab_bench.py --passes 1 lib_prof.so 2>&1 | grep "lane_prof" | head -8
PW_DEBUG_ROUNDS=1 timeout 300 python tools/ab_bench.py --passes 2 --scale 18 libpecanpy_amd.so 2>&1 | grep -v "^{" | tail -8
# partial index at RMAT-22: budget = 2 GB of lists (of 5.2)
PECANPY_AMD_INDEX_BUDGET=2000000000 PW_DEBUG_ROUNDS=1 timeout 600 python tools/ab_bench.py --passes 1 libpecanpy_amd.so > gpurun_out/c9/partial22.log 2>&1
python - <<'PY'
import json
for ln in open("gpurun_out/c9/partial22.log"):
    if ln.startswith("{"):
        d=json.loads(ln); p=d["passes"][-1]
        print("partial RMAT-22 (2 GB list budget): idx GB", d["index_GB"], "build", d["index_build_ms"], "ms", p["ms"], "rounds", p["rounds"], "ck", p["checksum"]%100000, "redo", p["redo"])
    elif "round" in ln and "round 0" in ln or "round 1:" in ln: print(ln.rstrip()[:100])
PY
