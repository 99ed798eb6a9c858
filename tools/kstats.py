#!/usr/bin/env python3
"""Per-step view of a rocprofv3 kernel_stats.csv: launches per step, average duration, time per step.   usage: tools/kstats.py file.csv steps [min_us]"""
import csv, re, sys
f, steps = sys.argv[1], int(sys.argv[2])
mn = float(sys.argv[3]) if len(sys.argv) > 3 else 15.0
rows = list(csv.DictReader(open(f)))
tot = n = 0.0
for r in rows:
    c = int(r["Calls"]) / steps; t = int(r["TotalDurationNs"]) / steps / 1e3
    tot += t; n += c
    nm = re.sub(r"\(.*", "", r["Name"])[:100]
    if t > mn:
        print("%6.1f x %7.1f us = %7.1f us/step  %s" % (c, float(r["AverageNs"]) / 1e3, t, nm))
print("total %.1f us/step, %.1f launches/step" % (tot, n))
