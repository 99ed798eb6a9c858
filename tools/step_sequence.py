"""Launch sequence of ONE step from a rocprofv3 kernel trace (csv): kernel names in start order with durations, between two AdamW launches.
usage: python tools/step_sequence.py <kernel_trace.csv> [anchor substring, default adamw_kernel]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
anchor = sys.argv[2] if len(sys.argv) > 2 else "adamw_kernel"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]["End_Timestamp"])
print("launches between the last two '%s': %d" % (anchor, b - a))
for r in rows[a + 1:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f us  +%7.2f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, r["Kernel_Name"][:120]))
