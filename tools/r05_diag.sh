#!/bin/bash
# usage: tools/r05_diag.sh <tag> <bench args...>   (GPU box) -- memory-path diagnostics of the lane kernel, round 5:
# vector-memory pipeline busy, address translation, memory-side request level / latency, one rocprofv3 --pmc pass per group
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for grp in "TA_BUSY_avr TA_TA_BUSY_sum GRBM_GUI_ACTIVE TA_TOTAL_WAVEFRONTS_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum GRBM_UTCL2_BUSY" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_LEVEL_sum GRBM_EA_BUSY" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCC_TAG_STALL_sum TCC_BUSY_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_32B_sum" \
           "TCP_TCP_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_TCR_TCP_STALL_CYCLES_sum"; do
  i=$((i+1))
  rm -rf /tmp/dg_$i
  timeout 300 rocprofv3 --kernel-trace --pmc $grp -d /tmp/dg_$i -o p -- python $R/bench.py "$@" --no-cpu-baseline --no-host-call > /tmp/dg_$i.log 2>&1
  python $R/tools/prof_summary.py /tmp/dg_$i/p_results.db $OUT/pass$i.txt > /dev/null 2>&1 || { echo "# group failed: $grp" > $OUT/pass$i.txt; tail -5 /tmp/dg_$i.log >> $OUT/pass$i.txt; }
done
grep -hE "walk_lanes_kernel|lanes_chain_kernel|group failed" $OUT/pass*.txt | grep -v "^void.*| [0-9]* | [0-9.]* | [0-9.]*$"
