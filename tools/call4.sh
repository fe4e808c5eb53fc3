#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/c4
export PYTHONUNBUFFERED=1
timeout 600 python tools/ab_bench.py --passes 1 lib_prof.so lib_profbase.so > gpurun_out/c4/prof.log 2>&1
timeout 600 python tools/ab_bench.py --passes 3 libpecanpy_amd.so lib_wide.so > gpurun_out/c4/ab.log 2>&1
grep "lane_prof" gpurun_out/c4/prof.log
python - <<'PY'
import json
for f in ("prof","ab"):
    print("==",f)
    for ln in open(f"gpurun_out/c4/{f}.log"):
        if ln.startswith("{"):
            d=json.loads(ln); ps=d["passes"][1:]
            print(d["lib"], "| ms", [p["ms"] for p in ps], "lane", [p["lane_ms"] for p in ps], "rounds", ps[-1]["rounds"], "probes/step", ps[-1]["probes"]/ps[-1]["steps"], "ck", [p["checksum"] % 100000 for p in d["passes"]])
PY
