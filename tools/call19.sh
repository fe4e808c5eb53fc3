#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
mkdir -p gpurun_out/c19
( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $OLDPWD/bench.py --config C5 --steps 2 --warmup 1 --no-cpu-baseline > $OLDPWD/gpurun_out/c19/bench.json 2> /tmp/kt.err; python $OLDPWD/tools/prof_summary.py /tmp/kt/kt_results.db $OLDPWD/gpurun_out/c19/kernel_trace_C5.txt > /dev/null 2>&1 )
head -16 gpurun_out/c19/kernel_trace_C5.txt | cut -c1-160
