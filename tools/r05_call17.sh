#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=gpurun_out/r5c17; mkdir -p $O
python tools/ab_bench.py --scale 22 --passes 2 --p 0.3 --q 1.7 lib_prev.so libpecanpy_amd.so > $O/ab.txt 2>&1
echo "## 3.0 0.37" >> $O/ab.txt
python tools/ab_bench.py --scale 22 --passes 2 --p 3.0 --q 0.37 lib_prev.so libpecanpy_amd.so >> $O/ab.txt 2>&1
python - <<'PY'
import json
for ln in open("gpurun_out/r5c17/ab.txt"):
    if ln.startswith("##"): print(ln.strip())
    elif ln.startswith("{"):
        d = json.loads(ln); ps = d["passes"][1:]
        print("  ", d["lib"], "ms", [p["ms"] for p in ps], "Msteps", [p["Msteps_s"] for p in ps], "amb", ps[0]["amb"], "ck", [p["checksum"] % 100000 for p in ps])
PY
timeout 500 python -m pytest tests/test_gpu_parity.py tests/test_exact_decision.py -m gpu -x -q 2>&1 | tail -2
timeout 500 python -m pytest tests/test_gpu_scale.py -m gpu -x -q -k "float_lane or c5" 2>&1 | tail -2
timeout 500 python -m pytest tests/test_gpu_lanes.py -m gpu -x -q 2>&1 | tail -2
