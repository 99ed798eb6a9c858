#!/bin/bash
# HBM traffic of the dominant kernel (FFN-up GEMM) inside bench.py: separate --pmc passes as the guide prescribes.
# usage (GPU box): tools/pmc_bench.sh gpurun_out/pmc_bench
R=$PWD; O=$R/$1; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT TCC_MISS TCC_REQ TCC_EA0_RDREQ" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE"; do
  n=$(echo $C | cut -d' ' -f1)
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/$n -- python $R/bench.py --steps 4 --warmup 2 --no-cpu --no-roofline --no-extra > $O/$n.log 2>&1
done
python $R/tools/pmc_summary.py $O
