#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counter CSVs (one counter per pass)."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main(root):
    for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
        acc = defaultdict(lambda: [0, 0.0])
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = (row.get("Kernel_Name", "?")[:70], row.get("Counter_Name", "?"))
                acc[k][0] += 1
                acc[k][1] += float(row.get("Counter_Value", 0) or 0)
        print("#", os.path.relpath(f, root))
        for (kn, cn), (n, s) in sorted(acc.items()):
            if "mrca" in kn:
                print(f"{kn:<72} {cn:<12} dispatches={n:<6} avg={s / n:.1f}")


def traffic_json(root, out, robots, scenario):
    """profiles/pmc_traffic.json: average FETCH_SIZE / WRITE_SIZE (KiB) per launch of the two env kernels, stamped with
    the hash of the kernel sources they were measured on (bench.py refuses a stale file)."""
    import json
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import bench
    acc = defaultdict(lambda: [0, 0.0])
    for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                kn, cn = row.get("Kernel_Name", ""), row.get("Counter_Name", "")
                for key in ("raycast_kernel", "move_kernel"):
                    if key in kn and cn in ("FETCH_SIZE", "WRITE_SIZE"):
                        acc[(key, cn)][0] += 1
                        acc[(key, cn)][1] += float(row.get("Counter_Value", 0) or 0)
    avg = {k: v[1] / v[0] for k, v in acc.items() if v[0]}
    d = {"source": os.path.basename(root.rstrip("/")), "robots": int(robots), "scenario": scenario,
         "kernel_src_sha16": bench.kernel_source_hash(),
         "fetch_kib": avg.get(("raycast_kernel", "FETCH_SIZE")), "write_kib": avg.get(("raycast_kernel", "WRITE_SIZE")),
         "move_fetch_kib": avg.get(("move_kernel", "FETCH_SIZE")), "move_write_kib": avg.get(("move_kernel", "WRITE_SIZE")),
         "dispatches": {f"{k[0]}:{k[1]}": acc[k][0] for k in acc},
         "note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), averages per launch in "
                 "KiB; FETCH_SIZE is doubled on gfx950 when converted to bytes (MI355X_MICROARCH.md HBM section)."}
    json.dump(d, open(out, "w"), indent=1)
    print(json.dumps(d))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "--traffic-json":
        traffic_json(sys.argv[1], sys.argv[3], sys.argv[4], sys.argv[5] if len(sys.argv) > 5 else "stage1")
    else:
        main(sys.argv[1])
