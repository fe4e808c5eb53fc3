#!/bin/bash
# usage: tools/ab_lines.sh <out dir under gpurun_out> -- headline A/B lines of round 6 (sampled verification on / off, self loops)
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$1; mkdir -p $OUT; cd $R
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-call"
timeout 300 $B > $OUT/headline_sample_on.json 2> $OUT/err1.txt
PECANPY_AMD_VERIFY_SAMPLE=0 timeout 300 $B > $OUT/headline_sample_off.json 2> $OUT/err2.txt
timeout 400 $B --self-loops 1000 > $OUT/headline_loops1000.json 2> $OUT/err3.txt
timeout 400 $B --self-loops 1000 --p 0.3 --q 1.7 --steps 2 --warmup 1 > $OUT/floats_loops1000.json 2> $OUT/err4.txt
for f in $OUT/*.json; do python - $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
    print(sys.argv[1].split('/')[-1], d['value'], 'ms/pass', d['ms_per_step'], 'lane ms', r['avg_launch_ms'], 'rounds', r.get('lane_rounds'), 'redo', r.get('redo_walks'), 'create', d['config']['graph_create_wall_ms'], 'index ms', d['config']['graph_index_build_ms'])
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
done
