#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/c5
export PYTHONUNBUFFERED=1
PECANPY_AMD_LIB=$PWD/pecanpy_amd/lib_wd.so timeout 300 python tools/lanes_check.py 14 18 > gpurun_out/c5/wd.log 2>&1
rc=$?; echo "wd rc=$rc" | tee -a gpurun_out/c5/summary.txt
grep -c "equal=True" gpurun_out/c5/wd.log; grep "equal=False\|watchdog\|Error" gpurun_out/c5/wd.log | head
if [ $rc -ne 124 ]; then
  PW_DEBUG_ROUNDS=1 timeout 900 python tools/ab_bench.py --passes 3 lib_th48.so libpecanpy_amd.so > gpurun_out/c5/ab.log 2>&1
  timeout 400 python tools/ab_bench.py --passes 2 --p 0.3 --q 1.7 lib_th48.so libpecanpy_amd.so > gpurun_out/c5/floats.log 2>&1
  timeout 300 python tools/ab_bench.py --passes 3 --scale 18 lib_th48.so libpecanpy_amd.so > gpurun_out/c5/c2.log 2>&1
  timeout 600 python -m pytest tests/test_gpu_lane_index.py tests/test_gpu_lanes.py tests/test_exact_decision.py -m gpu -x -q > gpurun_out/c5/t1.log 2>&1
  echo "t1 rc=$?" | tee -a gpurun_out/c5/summary.txt; tail -3 gpurun_out/c5/t1.log
fi
python - <<'PY'
import json
for f in ("ab","floats","c2"):
    print("==",f)
    for ln in open(f"gpurun_out/c5/{f}.log"):
        if ln.startswith("{"):
            d=json.loads(ln); ps=d["passes"][1:]
            print(d["lib"], "create", d["create_wall_ms"], "idx", d["index_build_ms"], "| ms", [p["ms"] for p in ps], "lane", [p["lane_ms"] for p in ps], "rounds", ps[-1]["rounds"], "probes/step", round(ps[-1]["probes"]/ps[-1]["steps"],2), "ck", [p["checksum"] % 100000 for p in d["passes"]], "Msteps/s", ps[-1]["Msteps_s"])
        elif "round" in ln: print(ln.rstrip()[:120])
PY
