python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -q -x -k "attention" 2>&1 | tail -2
run() { python bench.py --no-cpu --no-extra --no-sustained --workload $1 --steps 8 --warmup 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernel_ms_per_step']
print('%-4s %.1f pairs/s ms/step %.3f ' % ('$1', d['value'], d['ms_per_step']), {a: round(b,3) for a,b in k.items()})"; }
run gqa; run vcr; run gqa; run vcr
