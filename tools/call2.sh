#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/c2
export PYTHONUNBUFFERED=1
PECANPY_AMD_LIB=$PWD/pecanpy_amd/lib_wd.so timeout 300 python tools/lanes_check.py 14 18 > gpurun_out/c2/wd.log 2>&1
rc=$?; echo "wd rc=$rc" | tee -a gpurun_out/c2/summary.txt
grep -c "equal=True" gpurun_out/c2/wd.log; grep "equal=False\|watchdog" gpurun_out/c2/wd.log | head
if [ $rc -ne 124 ]; then
  timeout 900 python tools/ab_bench.py --passes 3 lib_th48.so libpecanpy_amd.so lib_p32.so lib_drawlds.so lib_tb8.so > gpurun_out/c2/ab.log 2>&1
  echo "ab rc=$?" | tee -a gpurun_out/c2/summary.txt
  PW_DEBUG_ROUNDS=1 timeout 300 python tools/ab_bench.py --passes 1 libpecanpy_amd.so lib_base.so > gpurun_out/c2/rounds.log 2>&1
  timeout 400 python tools/ab_bench.py --passes 2 --p 0.3 --q 1.7 lib_base.so lib_drawlds.so > gpurun_out/c2/floats.log 2>&1
  timeout 300 python tools/ab_bench.py --passes 3 --scale 18 lib_base.so libpecanpy_amd.so lib_drawlds.so > gpurun_out/c2/c2.log 2>&1
  timeout 300 python -m pytest tests/test_gpu_lane_index.py -m gpu -x -q -k "directed_entry" > gpurun_out/c2/t1.log 2>&1
  echo "t1 rc=$?" | tee -a gpurun_out/c2/summary.txt; tail -3 gpurun_out/c2/t1.log
fi
python - <<'PY'
import json,glob
for f in ("ab","rounds","floats","c2"):
    print("==",f)
    for ln in open(f"gpurun_out/c2/{f}.log"):
        if ln.startswith("{"):
            d=json.loads(ln); ps=d["passes"][1:]
            print(d["lib"], "create", d["create_wall_ms"], "idx", d["index_build_ms"], "| ms", [p["ms"] for p in ps], "lane", [p["lane_ms"] for p in ps], "rng", ps[-1]["rng_ms"], "rounds", ps[-1]["rounds"], "chain", ps[-1]["chain"], "ck", [p["checksum"] % 100000 for p in d["passes"]], "Msteps/s", ps[-1]["Msteps_s"], "lk", ps[-1]["lane_kernel"])
        else:
            print(ln.rstrip()[:200])
PY
