#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=gpurun_out/r5c5; mkdir -p $O
python tools/ab_bench.py --scale 22 --passes 3 lib_e0.so libpecanpy_amd.so > $O/ab22.txt 2>&1
cat $O/ab*.txt | cut -c1-1000
tools/pmc_run.sh r5c5/pmc --steps 1 --warmup 0 > /dev/null 2>&1
grep -h "walk_lanes_kernel<false\|lanes_chain" gpurun_out/r5c5/pmc/pass*.txt | cut -c1-200
