#!/bin/bash
# usage: tools/final_set.sh <out dir under gpurun_out> -- the round's final measurement set: bench lines of every config, a kernel
# trace of the headline (rocprofv3 --kernel-trace --stats), PMC passes of the headline / FLOATS / C5 (tools/pmc_all.sh)
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$1; mkdir -p $R/gpurun_out/$OUT; cd $R
tools/bench_all.sh gpurun_out/$OUT/bench > /dev/null 2>&1
export TMPDIR=/tmp
(cd /tmp && rm -rf /tmp/ktrace && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/ktrace -o k -- python $R/bench.py --steps 5 --warmup 0 --no-cpu-baseline --no-host-call > $R/gpurun_out/$OUT/ktrace_bench.json 2> /dev/null; python $R/tools/prof_summary.py /tmp/ktrace/k_results.db $R/gpurun_out/$OUT/kernel_trace.txt > /dev/null 2>&1)
tools/pmc_all.sh $OUT/pmc headline floats C5 C2 C3 C4 > /dev/null 2>&1
# weighted dense matrices (round 6: walk_dense_weighted_kernel): sizes, then a kernel trace + counter passes at N = 20 000
for n in 2000 8000 20000 40000; do timeout 300 python tools/dense_weighted_bench.py $n 2>&1 | grep '^{' >> gpurun_out/$OUT/dense_weighted_sizes.jsonl; done
tools/dense_weighted_prof.sh $OUT/dense_weighted_prof 20000 > /dev/null 2>&1
head -c 600 gpurun_out/$OUT/bench/headline.json; echo; cut -c1-220 gpurun_out/$OUT/bench/configs.jsonl; head -12 gpurun_out/$OUT/kernel_trace.txt
