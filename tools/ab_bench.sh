#!/bin/bash
# same-box A/B of bench.py under different cpt_set_tuning settings: tools/ab_bench.sh "0=13" "0=3" ...
for rep in 1 2; do
for t in "$@"; do
  python bench.py --no-cpu --no-io --steps 40 --tune "$t" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
k = d['kernel_ms_per_step']
print('tune %-10s ms/step %.4f  qkv %.1f ao %.1f up %.1f down %.1f  embed %.1f img %.1f head %.1f' % ('$t', d['ms_per_step'], k['gemm_qkv'] / 12 * 1e3, k['gemm_attn_out'] / 12 * 1e3, k['gemm_ffn_up'] / 12 * 1e3, k['gemm_ffn_down'] / 12 * 1e3, k['embed_ln'] * 1e3, k['img_proj'] * 1e3, k['head'] * 1e3))"
done
done
