#!/usr/bin/env python3
"""Turn a rocprofv3 rocpd sqlite result into a small text summary: per-kernel calls / total /
average duration (--kernel-trace --stats) and, when present, PMC counter sums per kernel (--pmc).
usage: prof_summary.py results.db [out.txt]"""
import sqlite3
import sys


def main(db, out=None):
    con = sqlite3.connect(db)
    cur = con.cursor()
    lines = [f"# rocprofv3 summary of {db}"]
    try:
        rows = list(cur.execute(
            "select name, count(*), sum(end - start), avg(end - start) from kernels group by name "
            "order by 3 desc limit 12"))
        lines.append("kernel | calls | total_ms | avg_ms")
        for name, calls, total, avg in rows:
            lines.append(f"{name[:100]} | {calls} | {total / 1e6:.3f} | {avg / 1e6:.3f}")
    except sqlite3.Error as e:
        lines.append(f"# no kernel trace: {e}")
    try:
        rows = list(cur.execute(
            "select name, counter_name, sum(counter_value), count(distinct dispatch_id) from pmc_events "
            "group by name, counter_name order by name, counter_name"))
        if rows:
            lines.append("")
            lines.append("kernel | counter | sum over dispatches | dispatches")
            for k, c, v, n in rows:
                if "at::native" in k or "rocclr" in k:
                    continue
                lines.append(f"{k[:70]} | {c} | {v:.6g} | {n}")
    except sqlite3.Error as e:
        lines.append(f"# no PMC data: {e}")
    text = "\n".join(lines) + "\n"
    if out:
        with open(out, "w") as f:
            f.write(text)
    print(text)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
