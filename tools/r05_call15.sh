#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=gpurun_out/r5c15; mkdir -p $O
for j in 1310720 2621440 5242880 10485760; do
  echo "## chains=1 jobs $j" >> $O/ab.txt
  PECANPY_AMD_LANE_CHAINS=1 python tools/ab_bench.py --scale 22 --passes 2 --jobs $j lib_cth8.so lib_cth12.so lib_cth16.so lib_cth20.so >> $O/ab.txt 2>&1
done
echo "## chains=0 jobs 10485760" >> $O/ab.txt
PECANPY_AMD_LANE_CHAINS=0 python tools/ab_bench.py --scale 22 --passes 2 --jobs 10485760 lib_cth20.so >> $O/ab.txt 2>&1
python - <<'PY'
import json
for ln in open("gpurun_out/r5c15/ab.txt"):
    if ln.startswith("##"): print(ln.strip())
    elif ln.startswith("{"):
        d = json.loads(ln); ps = d["passes"][1:]
        print("  ", d["lib"], "ms", [p["ms"] for p in ps], "lane", [p["lane_ms"] for p in ps], "rounds", ps[0]["rounds"], "ck", [p["checksum"] % 100000 for p in ps])
PY
