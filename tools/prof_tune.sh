#!/bin/bash
# rocprofv3 kernel table of the training step under a cpt_set_tuning setting (development library).   usage: tools/prof_tune.sh <tag> "<k=v>" [batch]
T=$1; V=$2; BS=${3:-32}; R=$PWD; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/${T}_prof
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_prof -- python $R/bench.py --steps 10 --warmup 3 --mode train --batch $BS --no-cpu --no-sustained --tune "$V" > /dev/null 2> $O/${T}_prof.log
find $O/${T}_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${T}_kernel_stats.csv
rm -rf $O/${T}_prof
cd $R
python - <<PY
import csv
rows = list(csv.DictReader(open("$O/${T}_kernel_stats.csv")))
for r in [q for q in rows if any(t in q["Name"] for t in ("scale_cast","ln_bwd","adamw","ce_rows"))]:
    print("%-70s %6d %8.2f" % (r["Name"][:70], int(r["Calls"]), float(r["AverageNs"]) / 1e3))
PY
