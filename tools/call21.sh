#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
mkdir -p gpurun_out/c21
bash tools/pmc_all.sh c21/pmc C5 > gpurun_out/c21/pmc_all.log 2>&1
cp profiles/r04_traffic.json gpurun_out/c21/r04_traffic.json
( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $OLDPWD/bench.py --config C5 --steps 3 --warmup 1 --no-cpu-baseline > $OLDPWD/gpurun_out/c21/kt_bench.json 2> /tmp/kt.err; python $OLDPWD/tools/prof_summary.py /tmp/kt/kt_results.db $OLDPWD/gpurun_out/c21/kernel_trace_C5.txt > /dev/null 2>&1 )
bash tools/bench_all.sh gpurun_out/c21/bench > /dev/null 2>&1
python - <<'PY'
import json
d=json.loads(open("gpurun_out/c21/bench/headline.json").read().strip().splitlines()[-1])
print("headline", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("traffic"), {k:v for k,v in d["config"].items() if "index" in k or "create" in k or "first" in k})
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d.get("vs_cpu_baseline"))
for ln in open("gpurun_out/c21/bench/configs.jsonl"):
    try:
        d=json.loads(ln); print(d["config"]["baseline_config"], d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["kernel"][:60], d["roofline"].get("traffic"), d["config"].get("param_index_build_ms"))
    except Exception as e: print("bad line", ln[:100])
t=json.load(open("profiles/r04_traffic.json"))["workloads"]["rmat20w_SparseOTF_p0.5_q2_ext_w10_l80_seed0"]; print(t["kernel_ms_under_pmc"], t["fetch_bytes"]/1e9, t["issue"])
PY
