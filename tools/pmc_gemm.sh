#!/bin/bash
# usage: tools/pmc_gemm.sh <shape> <variant> <outdir>   (GPU box; separate --pmc passes)
R=$PWD; S=$1; V=$2; O=$R/$3; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT" \
         "SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_UNALIGNED_STALL GRBM_GUI_ACTIVE" \
         "TCC_HIT TCC_MISS TCC_REQ TCC_EA0_RDREQ" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/p$i -- python $R/tools/gemm_prof.py $S $V > $O/p$i.log 2>&1
done
python - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$O/p*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if "gemm" in row["Kernel_Name"]:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, v in acc.items():
            print("%-28s n=%3d mean=%.4g" % (k, len(v), sum(v) / len(v)))
PY
