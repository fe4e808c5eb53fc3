#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=gpurun_out/r5c12; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_lane_index.py -m gpu -x -q 2>&1 | tail -3
python tools/ab_bench.py --scale 22 --passes 3 lib_nosort.so libpecanpy_amd.so > $O/ab22.txt 2>&1
python tools/ab_bench.py --scale 22 --passes 2 --p 0.25 --q 4 lib_nosort.so libpecanpy_amd.so > $O/ab22c3.txt 2>&1
python - <<'PY'
import json
for f in ("ab22", "ab22c3"):
    for ln in open(f"gpurun_out/r5c12/{f}.txt"):
        if ln.startswith("{"):
            d = json.loads(ln); ps = d["passes"][1:]
            print(f, d["lib"], "create", d["create_wall_ms"], "index ms", d["index_build_ms"], "ms", [p["ms"] for p in ps], "lane", [p["lane_ms"] for p in ps], "ck", [p["checksum"] % 100000 for p in ps])
PY
for f in test_gpu_parity test_gpu_lanes; do timeout 500 python -m pytest tests/$f.py -m gpu -x -q 2>&1 | tail -2; done
