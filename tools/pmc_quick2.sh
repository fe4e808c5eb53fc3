#!/bin/bash
# usage: tools/pmc_quick2.sh <bench args...>  -- issue / wait / cache counters of the lane kernel (3 passes)
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" \
           "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_WAVES" \
           "TA_TA_BUSY_sum GRBM_GUI_ACTIVE TCP_TOTAL_CACHE_ACCESSES_sum TCC_REQ_sum" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  rm -rf /tmp/pq
  timeout 400 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pq -o p -- python $R/bench.py "$@" --no-cpu-baseline --no-host-call > /tmp/pq.log 2>&1
  python $R/tools/prof_summary.py /tmp/pq/p_results.db | grep -E "walk_lanes|walk_kernel|lanes_chain"
done
grep -o '"effective_steps_per_pass": [0-9]*' /tmp/pq.log | tail -1
