#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export PYTHONUNBUFFERED=1
timeout 400 python -m pytest tests/test_gpu_lanes.py -m gpu -x -q -k "weighted" 2>&1 | tail -12 | cut -c1-250
PW_DEBUG_ROUNDS=1 timeout 400 python bench.py --config C5 --steps 3 --warmup 1 --no-cpu-baseline 2> /tmp/c5.err | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('C5', d['value'], d['ms_per_step'], r['kernel'][:50], d['config']['param_index_build_ms'])"
grep "lanes\]" /tmp/c5.err | tail -14
PECANPY_AMD_NO_WLANES=1 timeout 400 python bench.py --config C5 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C5 wave kernel', d['value'], d['ms_per_step'])"
