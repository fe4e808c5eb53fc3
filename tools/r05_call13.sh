#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
timeout 300 python -m pytest tests/test_gpu_lane_index.py -m gpu -x -q 2>&1 | tail -3
bash tools/r05_create.sh 2>&1 | grep -v "^W2026" | head -40
cd $R
for f in test_gpu_parity test_gpu_lanes test_gpu_scale test_gpu_verify; do timeout 500 python -m pytest tests/$f.py -m gpu -x -q 2>&1 | tail -2; done
