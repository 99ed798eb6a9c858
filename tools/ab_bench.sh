#!/bin/bash
# same-box A/B of bench.py under different cpt_set_tuning settings: tools/ab_bench.sh "0=13" "0=3" ...
for rep in 1 2; do
for t in "$@"; do
  python bench.py --no-cpu --steps 40 --tune "$t" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
k = d['kernel_ms_per_step']
print('tune %-8s ms/step %.4f  qkv %.3f ao %.3f up %.3f down %.3f attn %.3f ln %.3f' % ('$t', d['ms_per_step'], k['gemm_qkv'], k['gemm_attn_out'], k['gemm_ffn_up'], k['gemm_ffn_down'], k.get('attention', 0), k.get('layernorm', 0)))"
done
done
