#!/bin/bash
# usage: tools/pmc_run.sh <tag> <bench args...>   (run on the GPU box via gpurun)
# One rocprofv3 pass per counter group (PMC slots: SQ 8, TCC 4); only text summaries are kept
# (the sqlite results are too large for gpurun_out/).
set -u
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_ANY" \
           "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "FETCH_SIZE" \
           "WRITE_SIZE" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_LEVEL_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  timeout 400 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmc_$i -o p -- python $R/bench.py "$@" --no-cpu-baseline --no-host-call > $OUT/pass$i.log 2>&1
  python $R/tools/prof_summary.py /tmp/pmc_$i/p_results.db $OUT/pass$i.txt > /dev/null 2>&1
  grep -m1 '^{"metric"' $OUT/pass$i.log | cut -c1-1500 > $OUT/pass$i.bench.txt
  rm -f $OUT/pass$i.log
done
cat $OUT/pass*.txt
