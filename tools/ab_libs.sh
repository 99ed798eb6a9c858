#!/bin/bash
# A/B of developer builds of libcpt_hip.so on the bench workload (GPU box): tools/ab_libs.sh lib1.so lib2.so ...
for l in "$@"; do
  CPT_LIB_PATH=$l python bench.py --no-cpu --steps 30 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
k = d['kernel_ms_per_step']
print('%-36s ms/step %.4f  qkv %.1f ao %.1f up %.1f down %.1f (us per launch)' % ('$l', d['ms_per_step'], k['gemm_qkv']/12*1e3, k['gemm_attn_out']/12*1e3, k['gemm_ffn_up']/12*1e3, k['gemm_ffn_down']/12*1e3))"
done
