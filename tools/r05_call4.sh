#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=gpurun_out/r5c4; mkdir -p $O
python tools/ab_bench.py --scale 22 --passes 3 lib_qi3.so libpecanpy_amd.so lib_pq5.so lib_pi3.so > $O/ab22.txt 2>&1
PW_DEBUG_ROUNDS=1 python tools/ab_bench.py --scale 22 --passes 2 --jobs 5242880 libpecanpy_amd.so lib_pi3.so > $O/ab_shard.txt 2>&1
python tools/ab_bench.py --scale 18 --passes 3 libpecanpy_amd.so lib_pi3.so > $O/ab18.txt 2>&1
python tools/ab_bench.py --scale 20 --passes 3 lib_base.so libpecanpy_amd.so lib_pi3.so > $O/ab20.txt 2>&1
cat $O/ab*.txt | cut -c1-1000
for f in test_gpu_lanes test_gpu_parity test_gpu_scale test_gpu_verify; do timeout 400 python -m pytest tests/$f.py -m gpu -x -q 2>&1 | tail -2; done
