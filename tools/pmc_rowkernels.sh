#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the row kernels (layernorm_rows, ln_bwd, embed_ln, adamw, head_*, pad_cast / embed_pad) in
# separate rocprofv3 --pmc passes over (a) bench.py's hbm_kernels leg + forward and (b) a training step.   usage (GPU box):
#   tools/pmc_rowkernels.sh r04_pmc_rows   -> gpurun_out/r04_pmc_rows/summary.json
R=$PWD; O=$R/gpurun_out/$1; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/inf_$C -- python $R/bench.py --steps 3 --warmup 1 --no-cpu --no-extra --no-sustained > $O/inf_$C.log 2>&1
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/trn_$C -- python $R/bench.py --steps 3 --warmup 1 --no-cpu --no-sustained --mode train > $O/trn_$C.log 2>&1
done
cd $R
python - <<P
import csv, glob, json, collections
fam = lambda n: next((k for k in ("layernorm768_kernel", "layernorm_rows_kernel", "ln_bwd_kernel", "ln_bwd_reduce", "embed_pad_kernel", "embed_ln_kernel", "embed_bwd", "adamw_kernel", "head_rows_ln3", "head_finish", "pad_cast_kernel",
                                    "colsum_kernel", "reduce_partials", "ce_rows", "zero_segments", "dropout_rows", "b64_regions_kernel", "tail_rows_kernel", "tail_finish_kernel", "gelu_parts_kernel") if k in n), None)
res = {}
for leg in ("inf", "trn"):
    for C in ("FETCH_SIZE", "WRITE_SIZE"):
        acc = collections.defaultdict(list)
        for f in glob.glob("$O/%s_%s/**/*counter_collection.csv" % (leg, C), recursive=True):
            for r in csv.DictReader(open(f)):
                k = fam(r["Kernel_Name"])
                if k and r["Counter_Name"] == C:
                    acc[(k, r["Grid_Size"])].append(float(r["Counter_Value"]))
        for (k, g), v in acc.items():
            res.setdefault("%s/%s/grid%s" % (leg, k, g), {})[C + "_KB_mean_per_launch"] = {"mean": sum(v) / len(v), "launches": len(v)}
for k, d in res.items():
    f = d.get("FETCH_SIZE_KB_mean_per_launch", {}).get("mean"); w = d.get("WRITE_SIZE_KB_mean_per_launch", {}).get("mean")
    if f is not None and w is not None:
        d["traffic_MB (2 x FETCH + WRITE, MI355X_MICROARCH.md gfx950 correction)"] = round((2 * f + w) / 1024, 3)
json.dump(res, open("$O/summary.json", "w"), indent=1, sort_keys=True)
print(json.dumps(res, indent=1, sort_keys=True)[:6000])
P
find $O -name "*.csv" -size +1M -delete 2>/dev/null; find $O -name "*.db" -delete 2>/dev/null
