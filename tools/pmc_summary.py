#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs: per kernel name, launches and the mean /
total counter value.  Usage: pmc_summary.py DIR [DIR ...]   (measurement tool)."""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    for d in sys.argv[1:]:
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        print(f"# {d}: {len(files)} counter file(s)")
        agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
        for f in files:
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    name = row.get("Kernel_Name", "?")
                    short = name.split("(")[0][-90:]
                    grid = row.get("Grid_Size", "")
                    key = (short, grid)
                    rec = agg[key][row.get("Counter_Name", "?")]
                    rec[0] += 1
                    rec[1] += float(row.get("Counter_Value", 0) or 0)
        rows = []
        for (short, grid), ctrs in agg.items():
            for cname, (n, tot) in ctrs.items():
                rows.append((tot, short, grid, cname, n))
        rows.sort(reverse=True)
        for tot, short, grid, cname, n in rows[:60]:
            print(f"{cname:12s} n={n:4d} mean={tot / n:14.1f} total={tot:16.1f} grid={grid:>10s} {short}")


if __name__ == "__main__":
    main()
