run() { python bench.py --no-cpu --no-extra --no-sustained --steps 30 --warmup 10 --tune $1 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernel_ms_per_step']
print('%-18s ms/step %.4f  qkv %.1f ao %.1f up %.1f down %.1f (us per launch) sum %.1f' % ('$1', d['ms_per_step'], k['gemm_qkv']/12*1e3, k['gemm_attn_out']/12*1e3, k['gemm_ffn_up']/12*1e3, k['gemm_ffn_down']/12*1e3, (k['gemm_qkv']+k['gemm_attn_out']+k['gemm_ffn_up']+k['gemm_ffn_down'])/12*1e3))"; }
for r in 1 2; do
run 24=8
run 24=4
run 24=0
run 24=8,15=0
run 24=4,15=0
run 24=0,15=0
done
