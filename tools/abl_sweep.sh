#!/bin/bash
# Per-kernel time of the bf16 bench under the GEMM ablation bits (cpt_set_tuning key 1):
#  1 no operand DMA, 2 no fragment reads, 4 no MFMA, 8 no epilogue, 16 no epilogue math, 64 no epilogue stores,
#  128 no residual loads, 256 no slab writes.   usage (GPU box): tools/abl_sweep.sh "0 8 16 64 ..."
for a in $1; do
  python bench.py --no-cpu --no-check --steps 30 --warmup 5 --tune "1=$a" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
k = d['kernel_ms_per_step']
print('abl %-4s ms/step %.4f  qkv %.1f ao %.1f up %.1f down %.1f (us per launch)' % ('$a', d['ms_per_step'], k['gemm_qkv']/12*1e3, k['gemm_attn_out']/12*1e3, k['gemm_ffn_up']/12*1e3, k['gemm_ffn_down']/12*1e3))"
done
