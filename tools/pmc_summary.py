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


if __name__ == "__main__":
    main(sys.argv[1])
