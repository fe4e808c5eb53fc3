#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=gpurun_out/r5c10; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_lane_index.py -m gpu -x -q 2>&1 | tail -5
PECANPY_AMD_CREATE_DEBUG=1 python tools/ab_bench.py --scale 22 --passes 1 libpecanpy_amd.so > $O/ab22.txt 2>&1
PECANPY_AMD_INDEX_TWO_PASS=1 python tools/ab_bench.py --scale 22 --passes 1 libpecanpy_amd.so > $O/ab22_twopass.txt 2>&1
python - <<'PY'
import json
for f in ("ab22", "ab22_twopass"):
    for ln in open(f"gpurun_out/r5c10/{f}.txt"):
        if ln.startswith("{"):
            d = json.loads(ln); ps = d["passes"]
            print(f, "create", d["create_wall_ms"], "index ms", d["index_build_ms"], "GB", d["index_GB"], "ms", [p["ms"] for p in ps], "ck", [p["checksum"] % 100000 for p in ps])
PY
for f in test_gpu_parity test_gpu_lanes test_gpu_scale; do timeout 500 python -m pytest tests/$f.py -m gpu -x -q 2>&1 | tail -2; done
