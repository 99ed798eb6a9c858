run() { python bench.py --no-cpu --no-extra --no-sustained --workload $1 --steps 8 --warmup 3 --tune $2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernel_ms_per_step']
print('%-4s %-6s %.1f pairs/s ms/step %.3f ' % ('$1', '$2', d['value'], d['ms_per_step']), {a: round(b,3) for a,b in k.items()})"; }
run gqa 29=0; run gqa 29=1; run gqa 29=2; run gqa 29=0; run gqa 29=1; run gqa 29=2
run vcr 29=0; run vcr 29=1; run vcr 29=2
python -m pytest tests -m gpu -q -x 2>&1 | tail -4
