#!/usr/bin/env python3
"""Condense a tools/profile.sh output directory into a small text summary (kernel stats + per-dispatch PMC means)."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
KEY = "gfw_"
for f in sorted(glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True)):
    print("== kernel stats:", os.path.relpath(f, out))
    for row in csv.DictReader(open(f)):
        print("  %-90s calls=%s total_ns=%s avg_ns=%s pct=%s" % (row.get("Name", "")[:90], row.get("Calls"), row.get("TotalDurationNs"), row.get("AverageNs"), row.get("Percentage")))
for d in ["pmc%d" % i for i in range(1, 13)]:
    for f in sorted(glob.glob(os.path.join(out, d, "**", "*counter_collection.csv"), recursive=True)):
        acc = defaultdict(lambda: defaultdict(list))
        for row in csv.DictReader(open(f)):
            name = row.get("Kernel_Name", "")
            if KEY not in name:
                continue
            acc[name[:80]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        print("== counters:", os.path.relpath(f, out))
        for k, cs in acc.items():
            print("  kernel", k)
            for cn, vals in sorted(cs.items()):
                print("    %-28s mean/dispatch = %.4g   (n=%d)" % (cn, sum(vals) / len(vals), len(vals)))
