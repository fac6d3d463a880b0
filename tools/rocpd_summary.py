#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite result (kernel trace [+ PMC]) into a small text table."""
import sqlite3
import sys


def main(path, out=None):
    db = sqlite3.connect(path)
    cur = db.cursor()
    lines = [f"# rocprofv3 --kernel-trace --stats summary of {path.split('/')[-1]}",
             f"{'kernel':<60} {'calls':>7} {'total_us':>12} {'avg_us':>10} {'pct':>6}"]
    for name, calls, total, avg, pct in cur.execute(
            "select name,total_calls,total_duration,average,percentage from top_kernels order by total_duration desc limit 12"):
        short = name if len(name) <= 58 else name[:55] + "..."
        lines.append(f"{short:<60} {calls:>7} {total:>12.1f} {avg:>10.3f} {pct:>6.2f}")
    try:
        rows = list(cur.execute(
            "select k.name, e.counter_name, count(*), avg(e.counter_value), sum(e.counter_value) from pmc_events e "
            "join kernels k on e.dispatch_id = k.dispatch_id group by k.name, e.counter_name"))
    except Exception as ex:  # schema differs between versions
        rows = []
        lines.append(f"# (no PMC table readable: {ex})")
    if rows:
        lines.append("")
        lines.append(f"{'kernel':<48} {'counter':<16} {'dispatches':>10} {'avg':>16} {'sum':>18}")
        for k, c, n, a, s in rows:
            if "mrca" not in k:
                continue
            short = k if len(k) <= 46 else k[:43] + "..."
            lines.append(f"{short:<48} {c:<16} {n:>10} {a:>16.1f} {s:>18.1f}")
    text = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(text)
    print(text)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
