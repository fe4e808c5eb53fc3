#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/c10
export PYTHONUNBUFFERED=1
PECANPY_AMD_LIB=$PWD/pecanpy_amd/lib_wd.so timeout 300 python tools/lanes_check.py 14 18 > gpurun_out/c10/wd.log 2>&1
rc=$?; echo "wd rc=$rc"; grep -c "equal=True" gpurun_out/c10/wd.log; grep "equal=False\|watchdog\|Error" gpurun_out/c10/wd.log | head
if [ $rc -ne 124 ]; then
  timeout 900 python tools/ab_bench.py --passes 3 libpecanpy_amd.so lib_linelds.so > gpurun_out/c10/ab.log 2>&1
  timeout 400 python tools/ab_bench.py --passes 2 --p 0.3 --q 1.7 libpecanpy_amd.so lib_linelds.so > gpurun_out/c10/floats.log 2>&1
  timeout 300 python tools/ab_bench.py --passes 3 --scale 18 libpecanpy_amd.so lib_linelds.so > gpurun_out/c10/c2.log 2>&1
fi
PECANPY_AMD_INDEX_BUDGET=2000000000 PW_DEBUG_ROUNDS=1 timeout 600 python tools/ab_bench.py --passes 1 libpecanpy_amd.so > gpurun_out/c10/partial22.log 2>&1
timeout 300 python -m pytest tests/test_gpu_lane_index.py -m gpu -x -q -s -k "partial" 2>&1 | grep "partial\]\|passed\|failed" | head -3
python - <<'PY'
import json
for f in ("ab","floats","c2","partial22"):
    print("==",f)
    for ln in open(f"gpurun_out/c10/{f}.log"):
        if ln.startswith("{"):
            d=json.loads(ln); ps=d["passes"][1:]
            print(d["lib"], "idx", d["index_build_ms"], d["index_GB"], "| ms", [p["ms"] for p in ps], "lane", [p["lane_ms"] for p in ps], "rounds", ps[-1]["rounds"], "probes/step", round(ps[-1]["probes"]/ps[-1]["steps"],2), "ck", [p["checksum"] % 100000 for p in d["passes"]], "redo", ps[-1]["redo"])
        elif f == "partial22" and ("round 0" in ln or "round 1:" in ln or "round 5:" in ln): print(ln.rstrip()[:100])
PY
