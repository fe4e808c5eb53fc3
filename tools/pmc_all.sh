#!/bin/bash
# usage: tools/pmc_all.sh <out dir> [configs...]   (on the GPU box): rocprofv3 --pmc passes (tools/pmc_run.sh) of one
# bench.py pass per BASELINE config, folded into profiles/r06_traffic.json (tools/pmc_to_json.py); the summaries are
# kept under <out dir>/<config>/ and the JSON is copied next to them (gpurun merges only gpurun_out/).
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$1; shift
CFGS=${@:-headline C2 C3 C5 C4}
cd $R
for c in $CFGS; do
  case $c in
    headline) ARGS="--steps 1 --warmup 0"; KERN="walk_lanes_kernel+lanes_chain_kernel"; KEY="rmat22_SparseOTF_p0.5_q2_w10_l80_seed0";;
    C2) ARGS="--config C2 --steps 1 --warmup 0"; KERN="walk_lanes_kernel+lanes_chain_kernel"; KEY="rmat18_SparseOTF_p0.5_q2_w10_l80_seed0";;
    C3) ARGS="--config C3 --steps 1 --warmup 0"; KERN="walk_lanes_kernel+lanes_chain_kernel"; KEY="rmat22_SparseOTF_p0.25_q4_w10_l80_seed0";;
    C5) ARGS="--config C5 --steps 1 --warmup 0"; KERN="walk_lanes_kernel+lanes_eager_weighted_kernel"; KEY="rmat20w_SparseOTF_p0.5_q2_ext_w10_l80_seed0";;
    C4) ARGS="--config C4 --steps 1 --warmup 0"; KERN="walk_dense_fast_kernel"; KEY="er100000_DenseOTF_p0.5_q2_w10_l80_seed0";;
    floats) ARGS="--p 0.3 --q 1.7 --steps 1 --warmup 0"; KERN="walk_lanes_kernel"; KEY="rmat22_SparseOTF_p0.3_q1.7_w10_l80_seed0";;
  esac
  tools/pmc_run.sh $OUT/$c $ARGS > /dev/null 2>&1
  STEPS=$(python - <<PY
import json
try:
    print(json.loads(open("$R/gpurun_out/$OUT/$c/pass1.bench.txt").read().split('"effective_steps_per_pass": ')[1].split(',')[0]))
except Exception:
    print(0)
PY
)
  python tools/pmc_to_json.py gpurun_out/$OUT/$c "$KEY" "$KERN" "$STEPS" > gpurun_out/$OUT/$c/json.log 2>&1
done
cp profiles/r06_traffic.json gpurun_out/$OUT/r06_traffic.json
