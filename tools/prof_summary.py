#!/usr/bin/env python3
"""Turn a rocprofv3 rocpd sqlite result (--kernel-trace --stats) into a small text summary
(per-kernel calls / total / average duration, plus PMC counters when present)."""
import sqlite3
import sys


def main(db, out=None):
    con = sqlite3.connect(db)
    cur = con.cursor()
    lines = [f"# rocprofv3 kernel summary of {db}", "name | calls | total_ms | avg_ms | pct   (top_kernels view reports microseconds)"]
    for name, calls, total, avg, pct in cur.execute(
            "select name, total_calls, total_duration, average, percentage from top_kernels"):
        lines.append(f"{name[:110]} | {calls} | {total / 1e3:.3f} | {avg / 1e3:.3f} | {pct:.2f}")
    try:
        rows = list(cur.execute(
            "select k.name, p.name, sum(e.value), count(*) from rocpd_pmc_event e "
            "join rocpd_info_pmc p on e.pmc_id = p.id join kernels k on e.event_id = k.id "
            "group by k.name, p.name"))
        if rows:
            lines.append("")
            lines.append("kernel | counter | sum | dispatches")
            for k, c, v, n in rows:
                lines.append(f"{k[:80]} | {c} | {v:.6g} | {n}")
    except sqlite3.Error as e:  # no counters in this run
        lines.append(f"# (no PMC data: {e})")
    text = "\n".join(lines) + "\n"
    if out:
        with open(out, "w") as f:
            f.write(text)
    print(text)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
