#!/bin/bash
# Timing experiments of the attention backward kernels inside the training step (development library): rocprofv3 average of the kernel per setting.
#   usage: tools/abf_probe.sh <tag> "<tune>" "<tune>" ...      -> gpurun_out/<tag>_abf.txt
T=$1; shift; R=$PWD; O=$R/gpurun_out; mkdir -p $O; : > $O/${T}_abf.txt
cd /tmp && export TMPDIR=/tmp
for BS in 32 4; do
  for V in "$@"; do
    rm -rf /tmp/abf_prof
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abf_prof -- python $R/bench.py --steps 8 --warmup 2 --mode train --batch $BS --no-cpu --no-sustained --no-check --tune "$V" > /dev/null 2>&1
    f=$(find /tmp/abf_prof -name "*kernel_stats.csv" | head -1)
    python - "$f" "$BS" "$V" <<'PY' | tee -a $O/${T}_abf.txt
import csv, sys
f, bs, v = sys.argv[1:4]
out = []
for r in csv.DictReader(open(f)):
    if "attn_bwd" in r["Name"] or "attention_kernel" in r["Name"]:
        out.append("%s %.1f us" % (r["Name"].split("(")[0].replace("void cpt::", "")[:40], float(r["AverageNs"]) / 1e3))
print("batch %s tune %-12s: %s" % (bs, v, "; ".join(out)))
PY
  done
done
