#!/bin/bash
# GPU call 1 (round 4): deferral form of the lane kernel -- safety run with the watchdog build, A/B at RMAT-22, new parity tests
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/c1
export PYTHONUNBUFFERED=1
PECANPY_AMD_LIB=$PWD/pecanpy_amd/lib_wd.so timeout 300 python tools/lanes_check.py 14 18 > gpurun_out/c1/wd.log 2>&1
rc=$?; echo "wd rc=$rc" | tee -a gpurun_out/c1/summary.txt
tail -12 gpurun_out/c1/wd.log
if [ $rc -ne 124 ]; then
  timeout 700 python tools/ab_bench.py --passes 3 lib_base.so libpecanpy_amd.so lib_th48.so lib_th16.so > gpurun_out/c1/ab.log 2>&1
  echo "ab rc=$?" | tee -a gpurun_out/c1/summary.txt
  cat gpurun_out/c1/ab.log | cut -c1-1500
  timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_lane_index.py -m gpu -x -q -k "deep_offsets or directed_entry" > gpurun_out/c1/t1.log 2>&1
  echo "t1 rc=$?" | tee -a gpurun_out/c1/summary.txt; tail -5 gpurun_out/c1/t1.log
  timeout 500 python -m pytest tests/test_gpu_scale.py -m gpu -x -q -k "float_lane or deep_in_the_stream or c4_full_size_fast or oracle_prefix" > gpurun_out/c1/t2.log 2>&1
  echo "t2 rc=$?" | tee -a gpurun_out/c1/summary.txt; tail -8 gpurun_out/c1/t2.log
fi
