#!/bin/bash
# usage: tools/pmc_ta.sh <tag> <bench args...>  -- vector-memory pipeline counters of the walk kernel
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for grp in "TA_BUSY_avr TA_TA_BUSY_sum GRBM_GUI_ACTIVE" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_READ_sum" \
           "TA_FLAT_READ_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  timeout 400 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmc_$i -o p -- python $R/bench.py "$@" --no-cpu-baseline --no-host-call > /tmp/pmc_$i.log 2>&1
  python $R/tools/prof_summary.py /tmp/pmc_$i/p_results.db $OUT/pass$i.txt > /dev/null 2>&1 || tail -3 /tmp/pmc_$i.log > $OUT/pass$i.err
done
cat $OUT/pass*.txt | grep -E "walk_kernel" 
