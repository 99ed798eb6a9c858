"""Idle time between consecutive kernels of the bench loop, from a rocprofv3 --kernel-trace CSV.
python tools/gap_analysis.py <kernel_trace.csv> [skip_first_n]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))[skip:]
busy = sum(e - s for s, e, _ in ev)
span = ev[-1][1] - ev[0][0]
gaps = defaultdict(lambda: [0, 0])
for (s0, e0, n0), (s1, e1, n1) in zip(ev, ev[1:]):
    g = max(0, s1 - e0)
    if g < 200000:            # ignore host-side pauses between bench phases
        k = n1.split("<")[0].split("(")[0][-60:]
        gaps[k][0] += g
        gaps[k][1] += 1
tot_gap = sum(v[0] for v in gaps.values())
print("kernels %d  span %.3f ms  busy %.3f ms (%.1f%%)  short gaps %.3f ms (%.1f%%)" %
      (len(ev), span / 1e6, busy / 1e6, 100.0 * busy / span, tot_gap / 1e6, 100.0 * tot_gap / span))
for k, (g, n) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:12]:
    print("  gap before %-62s n=%6d  mean %.2f us" % (k, n, g / n / 1e3))
