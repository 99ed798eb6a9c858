#!/bin/bash
# start-skew sweep of the two-workgroups-per-CU GEMMs (cpt_set_tuning key 7); usage: tools/skew_sweep.sh "0 2052 ..."
for a in $1; do
  python bench.py --no-cpu --steps 30 --warmup 5 --tune "7=$a" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
k = d['kernel_ms_per_step']
print('skew %-5s ms/step %.4f  qkv %.1f ao %.1f up %.1f down %.1f (us per launch)' % ('$a', d['ms_per_step'], k['gemm_qkv']/12*1e3, k['gemm_attn_out']/12*1e3, k['gemm_ffn_up']/12*1e3, k['gemm_ffn_down']/12*1e3))"
done
