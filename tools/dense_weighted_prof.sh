#!/bin/bash
# usage: tools/dense_weighted_prof.sh <tag> [N=20000]   (run on the GPU box via gpurun)
# tools/dense_weighted_bench.py under rocprofv3: a kernel trace and one pass per counter group (text summaries only).
set -u
TAG=$1; N=${2:-20000}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/dwk
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/dwk -o k -- python $R/tools/dense_weighted_bench.py $N > $OUT/trace_bench.txt 2>&1
python $R/tools/prof_summary.py /tmp/dwk/k_results.db $OUT/kernel_trace.txt > /dev/null 2>&1
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_ANY" \
           "FETCH_SIZE" \
           "WRITE_SIZE" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  rm -rf /tmp/dwp_$i
  timeout 300 rocprofv3 --kernel-trace --pmc $grp -d /tmp/dwp_$i -o p -- python $R/tools/dense_weighted_bench.py $N > $OUT/pass$i.log 2>&1
  python $R/tools/prof_summary.py /tmp/dwp_$i/p_results.db $OUT/pass$i.txt > /dev/null 2>&1
  rm -f $OUT/pass$i.log
done
grep -h "dense_weighted\|^kernel" $OUT/kernel_trace.txt $OUT/pass*.txt
