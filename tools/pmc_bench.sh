#!/bin/bash
# HBM traffic of the dominant kernel (FFN-up GEMM) inside bench.py: separate --pmc passes as the guide prescribes.
# usage (GPU box): tools/pmc_bench.sh gpurun_out/pmc_bench
R=$PWD; O=$R/$1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT TCC_MISS TCC_REQ TCC_EA0_RDREQ"; do
  n=$(echo $C | cut -d' ' -f1)
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/$n -- python $R/bench.py --steps 4 --warmup 2 --no-cpu --no-roofline > $O/$n.log 2>&1
done
python - <<PY
import csv, glob, collections, json
res = {}
for f in glob.glob("$O/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        fam = None
        if "gemm_pipe_kernel" in k:
            if "Li1EDF16b" in k or ", 1, " in k: fam = "gemm_ffn_up(+gelu)"
            elif "Li3Ef" in k: fam = "gemm_resid(attn_out,ffn_down)"
            elif "Li0EDF16b" in k: fam = "gemm_qkv"
        elif "layernorm_rows" in k: fam = "layernorm"
        elif "attention_kernel" in k: fam = "attention"
        if fam: acc[fam][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for fam, d in acc.items():
        for c, v in d.items():
            res.setdefault(fam, {})[c] = {"mean_per_launch": sum(v) / len(v), "launches": len(v)}
print(json.dumps(res, indent=1))
open("$O/summary.json", "w").write(json.dumps(res, indent=1))
PY
