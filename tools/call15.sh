#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export PYTHONUNBUFFERED=1
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "in_parts or padding or deep_offsets" 2>&1 | tail -3
for pp in 1 2 4 8; do echo "parts $pp:"; PECANPY_AMD_PARTS=$pp PECANPY_AMD_COPY_DEBUG=1 timeout 200 python tools/host_path.py 18 2>&1 | grep "RMAT\|pw_simulate\]" | tail -2; done
echo "default:"; timeout 200 python tools/host_path.py 18 2>&1 | tail -1
echo "RMAT-20 default / 1 part:"; timeout 300 python tools/host_path.py 20 2>&1 | tail -1; PECANPY_AMD_PARTS=1 timeout 300 python tools/host_path.py 20 2>&1 | tail -1
timeout 300 python bench.py --config C5 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('C5', d['value'], r['frac'], r['declared_bytes_per_launch'], r['declared_format'][:90])"
timeout 300 python bench.py --config C4 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('C4', d['value'], r['frac'], r['kernel'], r.get('traffic'), r.get('traffic_raw_counter_bytes'))"
