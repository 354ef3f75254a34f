"""Summarise rocprofv3 --pmc counter_collection.csv files per kernel (mean per dispatch)."""
import csv
import glob
import os
import sys
from collections import defaultdict


def summarise(root):
    rows = defaultdict(lambda: defaultdict(list))
    for path in glob.glob(os.path.join(root, "**", "*_counter_collection.csv"), recursive=True):
        with open(path) as f:
            for r in csv.DictReader(f):
                name = r["Kernel_Name"].split("(")[0]
                rows[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = []
    for k, counters in sorted(rows.items()):
        for c, vals in sorted(counters.items()):
            out.append((k, c, len(vals), sum(vals) / len(vals), min(vals), max(vals)))
    return out


if __name__ == "__main__":
    for k, c, n, mean, lo, hi in summarise(sys.argv[1]):
        print(f"{k[:100]:100s} {c:24s} n={n:5d} mean={mean:16.1f} min={lo:16.1f} max={hi:16.1f}")
