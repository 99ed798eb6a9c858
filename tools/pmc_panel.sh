#!/bin/bash
# FETCH_SIZE / WRITE_SIZE per launch of the stand-alone producers (tools/panel_bench.py) for N = 768 and N = 192 (GPU box): tools/pmc_panel.sh
R=$PWD; O=$R/gpurun_out/pmc_panel; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for n in 768 192; do
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/${C}_$n -- python $R/tools/panel_bench.py --rounds 1 --iters 10 --n $n --no-cold > $O/${C}_$n.log 2>&1
  python - <<P
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$O/${C}_$n/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        fam = "panel" if "prod3_panel" in k else ("rowmajor" if "gemm_pipe_kernel" in k else None)
        if fam: acc[(fam, r["Grid_Size"], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print("N=$n", k, "mean per launch %.1f MB over %d launches" % (sum(v) / len(v) / 1024, len(v)))
P
done
done
