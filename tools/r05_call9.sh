#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=gpurun_out/r5c9; mkdir -p $O
for j in 1310720 2621440 5242880 0; do
    echo "## jobs $j" >> $O/ab_sizes.txt
    PW_DEBUG_ROUNDS=1 python tools/ab_bench.py --scale 22 --passes 3 --jobs $j libpecanpy_amd.so 2>&1 | grep -v "round [1-9]" >> $O/ab_sizes.txt
done
for sc in 18 20; do echo "## scale $sc" >> $O/ab_sizes.txt; python tools/ab_bench.py --scale $sc --passes 3 libpecanpy_amd.so >> $O/ab_sizes.txt 2>&1; done
echo "## floats" >> $O/ab_sizes.txt; python tools/ab_bench.py --scale 22 --passes 2 --p 0.3 --q 1.7 libpecanpy_amd.so >> $O/ab_sizes.txt 2>&1
python - <<'PY'
import json
for ln in open("gpurun_out/r5c9/ab_sizes.txt"):
    if ln.startswith("##"): print(ln.strip(), end="  ")
    elif ln.startswith("{"):
        d = json.loads(ln); ps = d["passes"][1:]
        print("ms", [p["ms"] for p in ps], "lane", [p["lane_ms"] for p in ps], "rounds", ps[0]["rounds"], "chain", ps[0]["chain"], "ck", [p["checksum"] % 100000 for p in ps])
PY
for f in test_gpu_parity test_gpu_lanes test_gpu_scale test_gpu_verify test_gpu_lane_index test_gpu_cli test_gpu_sharding; do timeout 500 python -m pytest tests/$f.py -m gpu -x -q 2>&1 | tail -2; done
