#!/bin/bash
# round 5, GPU call 1: A/B of existing build options at RMAT-22 + shard-sized call + memory-path diagnostics
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=gpurun_out/r5c1; mkdir -p $O
python tools/ab_bench.py --scale 22 --passes 3 lib_base.so lib_p32.so lib_dl32.so lib_w6p32.so > $O/ab_variants.txt 2>&1
PECANPY_AMD_LANE_OCC=4 python tools/ab_bench.py --scale 22 --passes 3 lib_base.so lib_dl32.so > $O/ab_occ4.txt 2>&1
PECANPY_AMD_LANE_OCC=3 python tools/ab_bench.py --scale 22 --passes 3 lib_base.so > $O/ab_occ3.txt 2>&1
PECANPY_AMD_LANE_TAILS=1 python tools/ab_bench.py --scale 22 --passes 3 lib_base.so > $O/ab_tails.txt 2>&1
PW_DEBUG_ROUNDS=1 python tools/ab_bench.py --scale 22 --passes 3 --jobs 5242880 lib_base.so lib_dl32.so > $O/ab_shard.txt 2>&1
PW_DEBUG_ROUNDS=1 python tools/ab_bench.py --scale 22 --passes 2 --p 0.3 --q 1.7 lib_base.so > $O/ab_floats.txt 2>&1
cat $O/ab_*.txt | cut -c1-900
tools/r05_diag.sh r5c1/diag --steps 1 --warmup 0
