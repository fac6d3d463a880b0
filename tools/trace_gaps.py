#!/usr/bin/env python3
"""Durations AND gaps of consecutive dispatches from a rocprofv3 --kernel-trace csv.

    python tools/trace_gaps.py <dir or *_kernel_trace.csv> [name-substring ...]

For every kernel whose name contains one of the substrings (default: all): calls, median / mean duration, and the
gap between its end and the start of the NEXT dispatch in the trace (median, mean, p10, p90) -- the device-side cost
of a launch boundary as the profiler's timestamps see it.  Also the steady-state period of repeating patterns
(start-to-start of consecutive dispatches of the same kernel)."""
import csv
import glob
import os
import statistics as st
import sys


def load(path):
    if os.path.isdir(path):
        hits = sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))
        if not hits:
            raise SystemExit(f"no *kernel_trace.csv under {path}")
        path = hits[0]
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    return path, rows


def short(name):
    name = name.replace("(anonymous namespace)::", "").split("(")[0]
    return name if len(name) <= 56 else "..." + name[-53:]


def pct(v, q):
    v = sorted(v)
    return v[min(len(v) - 1, int(q * len(v)))]


def main():
    if len(sys.argv) < 2:
        raise SystemExit(__doc__)
    path, rows = load(sys.argv[1])
    want = sys.argv[2:]
    by = {}
    for i, (s, e, n) in enumerate(rows):
        if want and not any(w in n for w in want):
            continue
        d = by.setdefault(n, {"dur": [], "gap": [], "next": {}, "starts": []})
        d["dur"].append((e - s) * 1e-3)
        d["starts"].append(s)
        if i + 1 < len(rows):
            d["gap"].append((rows[i + 1][0] - e) * 1e-3)
            nn = short(rows[i + 1][2])
            d["next"][nn] = d["next"].get(nn, 0) + 1
    print(f"# {path}: {len(rows)} dispatches; times in us; gap = start of the NEXT dispatch - end of this one")
    print(f"{'kernel':<58} {'calls':>6} {'dur med':>8} {'dur mean':>9} {'gap med':>8} {'gap mean':>9} {'gap p10':>8} {'gap p90':>8} "
          f"{'period med':>10}  next")
    for n, d in sorted(by.items(), key=lambda kv: -sum(kv[1]["dur"])):
        g = d["gap"] or [float("nan")]
        per = [(b - a) * 1e-3 for a, b in zip(d["starts"], d["starts"][1:])] or [float("nan")]
        nxt = max(d["next"].items(), key=lambda kv: kv[1])[0] if d["next"] else "-"
        print(f"{short(n):<58} {len(d['dur']):>6} {st.median(d['dur']):>8.2f} {st.mean(d['dur']):>9.2f} {st.median(g):>8.2f} "
              f"{st.mean(g):>9.2f} {pct(g, 0.1):>8.2f} {pct(g, 0.9):>8.2f} {st.median(per):>10.2f}  {nxt}")


if __name__ == "__main__":
    main()
