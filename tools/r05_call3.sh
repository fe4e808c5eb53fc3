#!/bin/bash
# round 5, GPU call 3: QUAD form -- correctness (GPU suite) + A/B against the round-4 build
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=gpurun_out/r5c3; mkdir -p $O
python tools/ab_bench.py --scale 22 --passes 3 lib_base.so libpecanpy_amd.so lib_q64.so lib_qi3.so > $O/ab22.txt 2>&1
PW_DEBUG_ROUNDS=1 python tools/ab_bench.py --scale 22 --passes 3 --jobs 5242880 lib_base.so libpecanpy_amd.so lib_qi3.so > $O/ab_shard.txt 2>&1
python tools/ab_bench.py --scale 18 --passes 3 lib_base.so libpecanpy_amd.so lib_qi3.so > $O/ab18.txt 2>&1
python tools/ab_bench.py --scale 22 --passes 2 --p 0.25 --q 4 lib_base.so libpecanpy_amd.so > $O/ab22_c3.txt 2>&1
cat $O/ab*.txt | cut -c1-1200
tools/gpu_suite.sh gpurun_out/r5c3/suite 400
