#!/bin/bash
# usage: tools/gpu_suite.sh <out dir> [per-file timeout s]   (run on the GPU box via gpurun)
# Every test file in its own pytest process with its own timeout: a hung kernel costs one file, not the call.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/$1; T=${2:-240}
mkdir -p $OUT
cd $R
for f in tests/test_*.py; do
  n=$(basename $f .py)
  s=$(date +%s)
  timeout $T python -m pytest $f -m gpu -x -q --durations=6 > $OUT/$n.log 2>&1
  rc=$?
  echo "$n rc=$rc $(( $(date +%s) - s ))s $(tail -1 $OUT/$n.log | cut -c1-120)" >> $OUT/summary.txt
done
cat $OUT/summary.txt
