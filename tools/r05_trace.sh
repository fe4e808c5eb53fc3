#!/bin/bash
# usage: tools/r05_trace.sh <tag> [bench args]  (GPU box): bench line + rocprofv3 kernel trace summary of the same command
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; O=$R/gpurun_out/$1; shift; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python bench.py --steps 5 --warmup 2 "$@" > $O/bench.json 2> $O/bench.err
cd /tmp; rm -rf /tmp/kt
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -o p -- python $R/bench.py --steps 5 --warmup 0 --no-cpu-baseline --no-host-call "$@" > $O/trace_bench.json 2> $O/trace.err
python $R/tools/prof_summary.py /tmp/kt/p_results.db $O/kernel_trace.txt > /dev/null 2>&1
python - <<PY
import sqlite3
con = sqlite3.connect("/tmp/kt/p_results.db")
rows = list(con.execute("select name, count(*), sum(end - start), avg(end - start) from kernels group by name order by 3 desc limit 40"))
with open("$O/kernel_trace_full.txt", "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats: python bench.py --steps 5 --warmup 0 $*\nkernel | calls | total_ms | avg_ms\n")
    for n, c, t, a in rows:
        if "at::native" in n: continue
        f.write(f"{n[:110]} | {c} | {t/1e6:.3f} | {a/1e6:.3f}\n")
print(open("$O/kernel_trace_full.txt").read())
PY
head -c 1500 $O/bench.json
