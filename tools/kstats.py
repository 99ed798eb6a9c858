"""Per-step table of a rocprofv3 kernel_stats.csv: python tools/kstats.py file.csv steps [top]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:top]:
    print("%-96s %7.1f/step %8.1f us  %7.3f ms/step" % (r["Name"][:96], int(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3,
                                                        float(r["TotalDurationNs"]) / 1e6 / steps))
print("total %.3f ms/step, %.0f launches/step" % (tot / 1e6 / steps, sum(int(r["Calls"]) for r in rows) / steps))
