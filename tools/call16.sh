#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
mkdir -p gpurun_out/c16
bash tools/pmc_all.sh c16/pmc C3 C5 C4 > gpurun_out/c16/pmc_all.log 2>&1
cp profiles/r04_traffic.json gpurun_out/c16/r04_traffic.json
python - <<'PY'
import json
d=json.load(open("profiles/r04_traffic.json"))["workloads"]
for k,v in d.items(): print(k, v["kernel_ms_under_pmc"], round(v["fetch_bytes"]/1e9,1), round(v["write_bytes"]/1e9,1), v["issue"]["valu_per_step"], v["issue"]["valu_util"], v["sectors_per_step"], v["note"][:30])
PY
