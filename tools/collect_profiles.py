#!/usr/bin/env python3
"""Copy the summaries of a tools/final_set.sh run (gpurun_out/<tag>/...) into profiles/ under the round's names: the bench lines,
the kernel trace, one text file per config with its rocprofv3 --pmc passes, the traffic JSON, the weighted dense sizes + counters.
usage: python tools/collect_profiles.py <tag> [round prefix, default r06] [note]"""
import glob
import os
import re
import shutil
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
rnd = sys.argv[2] if len(sys.argv) > 2 else "r06"
note = sys.argv[3] if len(sys.argv) > 3 else f"tools/final_set.sh {tag}"
src = os.path.join(REPO, "gpurun_out", tag)
dst = os.path.join(REPO, "profiles")


def passes(d):
    fs = sorted(glob.glob(os.path.join(d, "pass*.txt")), key=lambda f: int(re.findall(r"pass(\d+)", f)[-1]))
    return [f for f in fs if not f.endswith(".bench.txt")]


shutil.copy(os.path.join(src, "bench", "headline.json"), os.path.join(dst, f"{rnd}_bench_rmat22_final.json"))
shutil.copy(os.path.join(src, "bench", "configs.jsonl"), os.path.join(dst, f"{rnd}_configs.jsonl"))
with open(os.path.join(dst, f"{rnd}_rmat22_kernel_trace.txt"), "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 0 ({note}); per pass: divide by 5\n")
    f.write(open(os.path.join(src, "kernel_trace.txt")).read())
for cfg in sorted(os.listdir(os.path.join(src, "pmc"))):
    d = os.path.join(src, "pmc", cfg)
    if not os.path.isdir(d):
        continue
    with open(os.path.join(dst, f"{rnd}_pmc_{cfg}.txt"), "w") as f:
        f.write(f"# bench.py {cfg} --steps 1 --warmup 0 (floats: --p 0.3 --q 1.7 on RMAT-22), one rocprofv3 --pmc pass per counter group "
                f"(tools/pmc_all.sh -> tools/pmc_run.sh), {note}\n")
        for i, p in enumerate(passes(d), 1):
            f.write(f"## pass {i}\n" + open(p).read())
if os.path.exists(os.path.join(src, "pmc", f"{rnd}_traffic.json")):
    shutil.copy(os.path.join(src, "pmc", f"{rnd}_traffic.json"), os.path.join(dst, f"{rnd}_traffic.json"))
if os.path.exists(os.path.join(src, "dense_weighted_sizes.jsonl")):
    shutil.copy(os.path.join(src, "dense_weighted_sizes.jsonl"), os.path.join(dst, f"{rnd}_dense_weighted_sizes.jsonl"))
dw = os.path.join(src, "dense_weighted_prof")
if os.path.isdir(dw):
    with open(os.path.join(dst, f"{rnd}_pmc_dense_weighted.txt"), "w") as f:
        f.write(f"# tools/dense_weighted_prof.sh <tag> 20000 ({note}): kernel trace, then one rocprofv3 --pmc pass per counter group; "
                "3 dispatches of each kernel per run (tools/dense_weighted_bench.py)\n")
        f.write(open(os.path.join(dw, "kernel_trace.txt")).read())
        for i, p in enumerate(passes(dw), 1):
            f.write(f"## pass {i}\n" + open(p).read())
print("collected", tag)
