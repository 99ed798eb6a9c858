#!/bin/bash
# VERDICT r3 item 1, the diagnosis: hipBLASLt's winning kernel against the fused LayerNorm producer (gemm_prod.hip) at the FFN-down and
# attn-out shapes under rocprofv3 --pmc (separate passes per counter group) + package power (GPU box).
#   tools/kloop_diag.sh <outdir under gpurun_out>      -> <outdir>/summary.json, power_*.txt; tools/kloop_diag_md.py renders the .md
R=$PWD; O=$R/gpurun_out/$1; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters_available.txt 2>&1
G1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT"
G2="SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU"
G3="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES GRBM_GUI_ACTIVE"
G4="SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_WAIT_INST_ANY"
G5="TCC_HIT TCC_MISS TCC_REQ TCC_EA0_RDREQ"
G6="FETCH_SIZE"
G7="WRITE_SIZE"
G8="TCP_TCC_READ_REQ TCP_TOTAL_ACCESSES TA_BUSY_CYCLES"
for shape in 3 1; do
  sn=$([ $shape = 3 ] && echo ffn_down || echo attn_out)
  i=0
  for C in "$G1" "$G2" "$G3" "$G4" "$G5" "$G6" "$G7" "$G8"; do
    i=$((i+1))
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/hbl_${sn}_p$i -- $R/tools/yardstick.bin --loop $shape 30 > $O/hbl_${sn}_p$i.log 2>&1
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/ours_${sn}_p$i -- python $R/tools/kloop_ours.py --shape $sn --iters 30 > $O/ours_${sn}_p$i.log 2>&1
  done
  # kernel trace alone (durations un-perturbed by counters)
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/hbl_${sn}_kt -- $R/tools/yardstick.bin --loop $shape 100 > $O/hbl_${sn}_kt.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/ours_${sn}_kt -- python $R/tools/kloop_ours.py --shape $sn --iters 100 > $O/ours_${sn}_kt.log 2>&1
  # power + clock while each loops for 6 s
  ( for k in $(seq 1 40); do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "sclk|Package Power" | tr '\n' ' '; echo; sleep 0.2; done ) > $O/power_hbl_$sn.txt &
  $R/tools/yardstick.bin --loop $shape 30 6 > $O/hbl_${sn}_sustained.json 2>&1; wait
  ( for k in $(seq 1 60); do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "sclk|Package Power" | tr '\n' ' '; echo; sleep 0.2; done ) > $O/power_ours_$sn.txt &
  python $R/tools/kloop_ours.py --shape $sn --iters 30 --seconds 6 > $O/ours_${sn}_sustained.json 2>&1; wait
done
cd $R
python tools/kloop_diag_md.py $O > $O/kloop_vs_hipblaslt.md 2> $O/md.err
find $O -name "*.csv" -size +2M -delete 2>/dev/null
find $O -name "*.db" -delete 2>/dev/null
ls $O | head -80
