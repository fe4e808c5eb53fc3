#!/bin/bash
# round 5 final measurements (GPU box): suite, bench lines of every config, kernel trace, PMC passes
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; T=r5final2; mkdir -p gpurun_out/$T
tools/gpu_suite.sh gpurun_out/$T/suite 400 > /dev/null 2>&1
cat gpurun_out/$T/suite/summary.txt
tools/bench_all.sh gpurun_out/$T/bench > /dev/null 2>&1
head -c 600 gpurun_out/$T/bench/headline.json; echo
bash tools/r05_trace.sh $T/trace > /dev/null 2>&1
tools/pmc_all.sh $T/pmc headline > /dev/null 2>&1
cp profiles/r05_traffic.json gpurun_out/$T/r05_traffic.json 2>/dev/null
ls gpurun_out/$T gpurun_out/$T/pmc
