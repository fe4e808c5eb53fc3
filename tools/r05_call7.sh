#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=gpurun_out/r5c7; mkdir -p $O
for j in 1310720 2621440 5242880 10485760 20971520 0; do
  for c in 0 1; do
    echo "## jobs $j chains $c" >> $O/ab_sizes.txt
    PECANPY_AMD_LANE_CHAINS=$c PW_DEBUG_ROUNDS=1 python tools/ab_bench.py --scale 22 --passes 2 --jobs $j libpecanpy_amd.so 2>&1 | grep -v "round [1-9]\|^#   \[lanes\] round 0.*round" >> $O/ab_sizes.txt
  done
done
for sc in 18 20; do for c in 0 1; do echo "## scale $sc chains $c" >> $O/ab_scales.txt; PECANPY_AMD_LANE_CHAINS=$c python tools/ab_bench.py --scale $sc --passes 3 libpecanpy_amd.so >> $O/ab_scales.txt 2>&1; done; done
python - <<'PY'
import json
for f in ("gpurun_out/r5c7/ab_sizes.txt", "gpurun_out/r5c7/ab_scales.txt"):
    for ln in open(f):
        if ln.startswith("##"): print(ln.strip(), end="  ")
        elif ln.startswith("{"):
            d = json.loads(ln); ps = d["passes"][1:]
            print("ms", [p["ms"] for p in ps], "lane", [p["lane_ms"] for p in ps], "rounds", ps[0]["rounds"], "chain", ps[0]["chain"], "ck", [p["checksum"] % 100000 for p in ps])
PY
for f in test_gpu_parity test_gpu_lanes test_gpu_scale test_gpu_verify test_gpu_lane_index; do timeout 500 python -m pytest tests/$f.py -m gpu -x -q 2>&1 | tail -2; done
