#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=gpurun_out/r5c16; mkdir -p $O
python tools/ab_bench.py --scale 22 --passes 3 lib_prev.so libpecanpy_amd.so > $O/ab.txt 2>&1
echo "## 5.2M" >> $O/ab.txt
python tools/ab_bench.py --scale 22 --passes 3 --jobs 5242880 lib_prev.so libpecanpy_amd.so >> $O/ab.txt 2>&1
python - <<'PY'
import json
for ln in open("gpurun_out/r5c16/ab.txt"):
    if ln.startswith("##"): print(ln.strip())
    elif ln.startswith("{"):
        d = json.loads(ln); ps = d["passes"][1:]
        print("  ", d["lib"], "ms", [p["ms"] for p in ps], "lane", [p["lane_ms"] for p in ps], "ck", [p["checksum"] % 100000 for p in ps])
PY
for f in test_gpu_parity test_gpu_lanes test_gpu_scale test_gpu_verify; do timeout 500 python -m pytest tests/$f.py -m gpu -x -q 2>&1 | tail -2; done
