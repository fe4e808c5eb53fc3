#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
mkdir -p gpurun_out/c14
# 1. kernel trace + stats of the driver's command shape (short)
( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $OLDPWD/bench.py --steps 4 --warmup 1 --no-cpu-baseline > $OLDPWD/gpurun_out/c14/kt_bench.json 2> /tmp/kt.err; python $OLDPWD/tools/prof_summary.py /tmp/kt/kt_results.db $OLDPWD/gpurun_out/c14/kernel_trace.txt > /dev/null 2>&1 )
head -14 gpurun_out/c14/kernel_trace.txt
# 2. PMC passes for headline + C2
bash tools/pmc_all.sh c14/pmc headline C2 > gpurun_out/c14/pmc_all.log 2>&1
cp profiles/r04_traffic.json gpurun_out/c14/r04_traffic.json
python - <<'PY'
import json
d=json.load(open("profiles/r04_traffic.json"))["workloads"]
for k in ("rmat22_SparseOTF_p0.5_q2_w10_l80_seed0","rmat18_SparseOTF_p0.5_q2_w10_l80_seed0"):
    v=d[k]; print(k, v["kernel_ms_under_pmc"], v["fetch_bytes"]/1e9, v["write_bytes"]/1e9, v["issue"], v["sectors_per_step"], v["note"][:40])
PY
# 3. bench lines of all configs
bash tools/bench_all.sh gpurun_out/c14/bench > /dev/null 2>&1
python - <<'PY'
import json
d=json.loads(open("gpurun_out/c14/bench/headline.json").read().strip().splitlines()[-1])
print("headline", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("traffic"), {k:v for k,v in d["config"].items() if "index" in k or "create" in k or "first" in k})
for ln in open("gpurun_out/c14/bench/configs.jsonl"):
    try:
        d=json.loads(ln); print(d["config"]["baseline_config"], d["config"]["workload"][:60], d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel"][:40])
    except Exception as e: print("bad line", ln[:100])
PY
