#!/usr/bin/env python3
"""A window of a rocprofv3 --kernel-trace csv as a timeline: start / end of every dispatch relative to the window's first start,
the queue it ran on and the workgroups it had -- to SEE whether two launches overlapped (bench.py --chains).

    python tools/trace_timeline.py <dir or *_kernel_trace.csv> [skip [count]]
"""
import csv
import glob
import os
import sys


def main():
    path = sys.argv[1]
    skip = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    count = int(sys.argv[3]) if len(sys.argv) > 3 else 48
    if os.path.isdir(path):
        path = sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))[0]
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"),
                         r.get("Grid_Size", r.get("Grid_Size_X", "?")), r.get("Workgroup_Size", r.get("Workgroup_Size_X", "?"))))
    rows.sort()
    win = rows[skip: skip + count]
    t0 = win[0][0]
    print(f"# {path}: dispatches {skip} .. {skip + len(win)} of {len(rows)}; us from the first start of the window")
    busy_until = 0
    for s, e, n, q, g, w in win:
        name = n.replace("(anonymous namespace)::", "").split("(")[0].replace("void mrca::", "")[:40]
        over = "  (starts %.2f us before the previous dispatch ended)" % ((busy_until - s) * 1e-3) if s < busy_until else ""
        print(f"{(s - t0) * 1e-3:9.2f} -> {(e - t0) * 1e-3:9.2f}  {(e - s) * 1e-3:7.2f} us  queue {q:>3}  grid {g:>8}  {name}{over}")
        busy_until = max(busy_until, e)
    span = (win[-1][1] - win[0][0]) * 1e-3
    ray = [r for r in win if "raycast" in r[2]]
    if len(ray) > 2:
        print(f"# window {span:.1f} us; ray-cast launches {len(ray)}; mean start-to-start {(ray[-1][0] - ray[0][0]) * 1e-3 / (len(ray) - 1):.2f} us")


if __name__ == "__main__":
    main()
