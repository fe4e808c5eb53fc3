#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export PYTHONUNBUFFERED=1
timeout 300 python tools/lanes_check.py 14 18 > /tmp/check.log 2>&1; echo "check rc=$? equal=$(grep -c 'equal=True' /tmp/check.log)"
PW_DEBUG_ROUNDS=1 timeout 900 python tools/ab_bench.py --passes 3 lib_chainnotail.so libpecanpy_amd.so lib_chain6.so 2>&1 | python -c "
import sys,json
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); ps=d['passes'][1:]; print(d['lib'],'ms',[p['ms'] for p in ps],'lane',[p['lane_ms'] for p in ps],'rng',ps[-1]['rng_ms'],'ck',[p['checksum']%100000 for p in d['passes']])
"
