#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=gpurun_out/r5c14; mkdir -p $O
for j in 2621440 5242880; do
  echo "## chains=1 jobs $j" >> $O/ab.txt
  PECANPY_AMD_LANE_CHAINS=1 python tools/ab_bench.py --scale 22 --passes 3 --jobs $j libpecanpy_amd.so lib_c4.so lib_cth20.so lib_cth36.so >> $O/ab.txt 2>&1
done
echo "## full size" >> $O/ab.txt
python tools/ab_bench.py --scale 22 --passes 3 libpecanpy_amd.so lib_dth40.so >> $O/ab.txt 2>&1
python - <<'PY'
import json
for ln in open("gpurun_out/r5c14/ab.txt"):
    if ln.startswith("##"): print(ln.strip())
    elif ln.startswith("{"):
        d = json.loads(ln); ps = d["passes"][1:]
        print("  ", d["lib"], "ms", [p["ms"] for p in ps], "lane", [p["lane_ms"] for p in ps], "rounds", ps[0]["rounds"], "chain", ps[0]["chain"], "ck", [p["checksum"] % 100000 for p in ps])
PY
