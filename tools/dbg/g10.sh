python -m pytest tests/test_gpu_ops.py -q -x -k "consumer or two_pass or ln_cons" 2>&1 | tail -4
run() { python bench.py --no-cpu --no-extra --no-sustained --steps 30 --warmup 10 --tune $1 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernel_ms_per_step']
print('%-8s ms/step %.4f  qkv %.1f ao %.1f up %.1f down %.1f (us per launch) sum %.1f' % ('$1', d['ms_per_step'], k['gemm_qkv']/12*1e3, k['gemm_attn_out']/12*1e3, k['gemm_ffn_up']/12*1e3, k['gemm_ffn_down']/12*1e3, (k['gemm_qkv']+k['gemm_attn_out']+k['gemm_ffn_up']+k['gemm_ffn_down'])/12*1e3))"; }
run 29=0; run 29=1; run 29=0; run 29=1
python bench.py --no-cpu --no-extra --no-sustained --workload gqa --steps 8 --warmup 3 --tune 29=0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('gqa 29=0', d['value'], d['ms_per_step'], {a: round(b,3) for a,b in d['kernel_ms_per_step'].items()})"
python bench.py --no-cpu --no-extra --no-sustained --workload gqa --steps 8 --warmup 3 --tune 29=1 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('gqa 29=1', d['value'], d['ms_per_step'], {a: round(b,3) for a,b in d['kernel_ms_per_step'].items()})"
