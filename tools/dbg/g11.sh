python -m pytest tests/test_gpu_ops.py -q -x -k "panel" 2>&1 | tail -2
for w in 8 4; do echo "== waves $w"; python tools/panel_bench.py --rounds 1 --iters 100 --no-cold --tune 24=$w 2>&1 | grep -E "panel kernel abl 0" | cut -c1-250; done
run() { python bench.py --no-cpu --no-extra --no-sustained --steps 30 --warmup 10 --tune $1 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernel_ms_per_step']
print('%-8s ms/step %.4f  qkv %.1f ao %.1f up %.1f down %.1f (us per launch) sum %.1f' % ('$1', d['ms_per_step'], k['gemm_qkv']/12*1e3, k['gemm_attn_out']/12*1e3, k['gemm_ffn_up']/12*1e3, k['gemm_ffn_down']/12*1e3, (k['gemm_qkv']+k['gemm_attn_out']+k['gemm_ffn_up']+k['gemm_ffn_down'])/12*1e3))"; }
run 24=8; run 24=0; run 24=4; run 24=8; run 24=0; run 24=4
