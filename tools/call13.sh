#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export PYTHONUNBUFFERED=1
PECANPY_AMD_LIB=$PWD/pecanpy_amd/lib_wd.so timeout 300 python tools/lanes_check.py 14 18 > /tmp/check.log 2>&1; rc=$?; echo "wd rc=$rc equal=$(grep -c 'equal=True' /tmp/check.log)"; grep "equal=False\|watchdog" /tmp/check.log | head -5
if [ $rc -ne 124 ]; then
timeout 900 python tools/ab_bench.py --passes 3 libpecanpy_amd.so lib_deep400.so lib_deep168.so lib_deep60.so 2>&1 | python -c "
import sys,json
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); ps=d['passes'][1:]; print(d['lib'],'ms',[p['ms'] for p in ps],'lane',[p['lane_ms'] for p in ps],'rounds',ps[-1]['rounds'],'chain',ps[-1]['chain'],'ck',[p['checksum']%100000 for p in d['passes']])
    else: print(ln.rstrip()[:200])
"
timeout 300 python tools/ab_bench.py --passes 3 --scale 18 libpecanpy_amd.so lib_deep168.so 2>&1 | python -c "
import sys,json
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); ps=d['passes'][1:]; print(d['lib'],'ms',[p['ms'] for p in ps],'lane',[p['lane_ms'] for p in ps],'ck',[p['checksum']%100000 for p in d['passes']])
"
fi
