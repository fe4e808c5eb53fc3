#!/usr/bin/env python3
"""Kernel timeline of the LAST pass in a rocprofv3 --kernel-trace result (rocpd sqlite): every dispatch from the last
mt_jump / mt_expand group on, with its duration and the gap to the dispatch before it -- what the serial parts of a pass
(jump-ahead tree, expansion, lane rounds, chain launches, host read-backs between them) cost one by one.
usage: pass_timeline.py results.db [out.txt]"""
import sqlite3
import sys


def main(db, out=None):
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select name, start, end from kernels order by start"))
    rows = [r for r in rows if "at::native" not in r[0]]
    # the last pass starts at the first mt_jump (or mt_expand) kernel behind the last walk kernel before it
    last = max(i for i, r in enumerate(rows) if "mt_expand_kernel" in r[0])
    first = last
    while first > 0 and ("mt_jump" in rows[first - 1][0] or "fillBuffer" in rows[first - 1][0]):
        first -= 1
    lines = [f"# last pass of {db}: kernel | duration us | gap to the previous dispatch us"]
    prev_end = rows[first][1]
    t0 = rows[first][1]
    for name, s, e in rows[first:]:
        lines.append(f"{name.split('(')[0][-60:]:60s} | {(e - s) / 1e3:9.1f} | {(s - prev_end) / 1e3:8.1f} | at {(s - t0) / 1e6:8.3f} ms")
        prev_end = e
    lines.append(f"# span {(prev_end - t0) / 1e6:.3f} ms")
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    print(text)


if __name__ == "__main__":
    main(*sys.argv[1:3])
