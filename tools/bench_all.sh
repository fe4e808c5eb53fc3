#!/bin/bash
# usage: tools/bench_all.sh <out dir>   (on the GPU box): one bench.py line per BASELINE config + the non-dyadic case
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/$1
mkdir -p $OUT
cd $R
: > $OUT/configs.jsonl
timeout 200 python bench.py --steps 5 --warmup 2 > $OUT/headline.json 2> $OUT/headline.err
for c in C2 C3 C5 C4; do
  timeout 300 python bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline --no-host-call 2> $OUT/$c.err | tail -1 >> $OUT/configs.jsonl
done
# unit graph, p and q not powers of two: no lane kernel (the float32 chain is the only decision), wave kernel with masks from the lists
timeout 300 python bench.py --p 0.3 --q 1.7 --steps 2 --warmup 1 --no-cpu-baseline --no-host-call 2> $OUT/nondyadic.err | tail -1 >> $OUT/configs.jsonl
# round 6: RMAT-22 with 1000 random self loops (the lane kernel keeps such graphs)
timeout 300 python bench.py --self-loops 1000 --steps 3 --warmup 1 --no-cpu-baseline --no-host-call 2> $OUT/loops.err | tail -1 >> $OUT/configs.jsonl
