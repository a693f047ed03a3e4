#!/usr/bin/env python
"""Per-kernel averages of a rocprofv3 --pmc run written with --output-format csv:  python tools/pmc_summary.py <dir> [kernel substring]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    root, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
    acc, cnt = defaultdict(float), defaultdict(int)
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name", "")
            if pat and pat not in k:
                continue
            key = (k.split("(")[0][:60], r["Counter_Name"])
            acc[key] += float(r["Counter_Value"])
            cnt[key] += 1
    for (k, c) in sorted(acc):
        print(f"{k:62s} {c:24s} {acc[(k, c)] / cnt[(k, c)]:14.4g}  (n={cnt[(k, c)]})")


if __name__ == "__main__":
    main()
