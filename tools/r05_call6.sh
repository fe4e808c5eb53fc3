#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=gpurun_out/r5c6; mkdir -p $O
python tools/ab_bench.py --scale 22 --passes 2 --p 0.3 --q 1.7 lib_base.so libpecanpy_amd.so lib_fw8.so lib_fw40.so > $O/ab22_floats.txt 2>&1
python tools/ab_bench.py --scale 22 --passes 2 --p 3.0 --q 0.37 libpecanpy_amd.so > $O/ab22_floats2.txt 2>&1
python tools/ab_bench.py --scale 22 --passes 3 lib_e0.so libpecanpy_amd.so > $O/ab22.txt 2>&1
cat $O/ab*.txt | cut -c1-1000
for f in test_exact_decision test_gpu_parity test_gpu_lanes test_gpu_scale; do timeout 500 python -m pytest tests/$f.py -m gpu -x -q 2>&1 | tail -3; done
