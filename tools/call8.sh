#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/c8
export PYTHONUNBUFFERED=1
timeout 400 python -m pytest tests/test_gpu_lane_index.py -m gpu -x -q -s -k "partial" > gpurun_out/c8/partial.log 2>&1; echo "partial rc=$?"; tail -12 gpurun_out/c8/partial.log | cut -c1-300
bash tools/gpu_suite.sh gpurun_out/c8/suite 400
