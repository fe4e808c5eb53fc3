#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/c11
export PYTHONUNBUFFERED=1
timeout 300 python tools/lanes_check.py 14 18 > gpurun_out/c11/check.log 2>&1; echo "check rc=$? equal=$(grep -c 'equal=True' gpurun_out/c11/check.log)"
for s in 18 20; do
  for t in 0 1; do
    echo "== scale $s tails $t"; PECANPY_AMD_LANE_TAILS=$t timeout 300 python tools/ab_bench.py --passes 3 --scale $s libpecanpy_amd.so 2>&1 | python -c "
import sys,json
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); ps=d['passes'][1:]; print('ms',[p['ms'] for p in ps],'lane',[p['lane_ms'] for p in ps],'rng',ps[-1]['rng_ms'],'Msteps/s',ps[-1]['Msteps_s'],'ck',[p['checksum']%100000 for p in d['passes']])
"
  done
done
for gsz in 256 512 2048 4096; do
  echo "== scale 18 gens $gsz"; PECANPY_AMD_MT_GENS=$gsz timeout 300 python tools/ab_bench.py --passes 3 --scale 18 libpecanpy_amd.so 2>&1 | python -c "
import sys,json
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); ps=d['passes'][1:]; print('ms',[p['ms'] for p in ps],'rng',[p['rng_ms'] for p in ps],'ck',[p['checksum']%100000 for p in d['passes']])
"
done
for gsz in 2048 4096; do
  echo "== scale 22 gens $gsz"; PECANPY_AMD_MT_GENS=$gsz timeout 300 python tools/ab_bench.py --passes 2 libpecanpy_amd.so 2>&1 | python -c "
import sys,json
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); ps=d['passes'][1:]; print('ms',[p['ms'] for p in ps],'rng',[p['rng_ms'] for p in ps],'ck',[p['checksum']%100000 for p in d['passes']])
"
done
