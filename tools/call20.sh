#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out; : > $R/gpurun_out/c20.txt; cd /tmp
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAVES SQ_BUSY_CYCLES" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  rm -rf /tmp/pq
  timeout 400 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pq -o p -- python $R/bench.py --config C5 --steps 1 --warmup 0 --no-cpu-baseline > /tmp/pq.log 2>&1
  python $R/tools/prof_summary.py /tmp/pq/p_results.db | grep -E "eager_weighted|walk_lanes_kernel" >> $R/gpurun_out/c20.txt
done
