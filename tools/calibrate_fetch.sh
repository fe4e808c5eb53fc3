#!/bin/bash
# usage (on the GPU box, via gpurun): tools/calibrate_fetch.sh [GiB] [iters]
# Random-gather throughput of the GPU + calibration of rocprofv3's FETCH_SIZE for scattered S-byte reads
# (the walk kernels' access pattern): tools/gather_bench issues a known number of accesses per kernel.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/gather
mkdir -p $OUT
GIB=${1:-8}; IT=${2:-64}
export TMPDIR=/tmp
cd /tmp
$R/tools/gather_bench $GIB $IT > $OUT/bench.jsonl 2> $OUT/bench.err
for c in FETCH_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_MISS_sum TCC_REQ_sum"; do
  tag=$(echo $c | cut -d' ' -f1)
  rm -rf /tmp/gb_$tag
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d /tmp/gb_$tag -o p -- $R/tools/gather_bench $GIB $IT > /dev/null 2> $OUT/pmc_$tag.err
  python $R/tools/prof_summary.py /tmp/gb_$tag/p_results.db $OUT/pmc_$tag.txt > /dev/null 2>&1
done
cat $OUT/bench.jsonl
cat $OUT/pmc_*.txt
