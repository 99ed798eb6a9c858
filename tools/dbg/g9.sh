run() { python bench.py --no-cpu --no-extra --no-sustained --workload gqa --steps 8 --warmup 3 --tune $1 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernel_ms_per_step']
print('%-14s %.1f pairs/s ms/step %.3f ' % ('$1', d['value'], d['ms_per_step']), {a: round(b,3) for a,b in k.items()})"; }
run 28=0,24=0
run 28=1,24=0
run 28=1,24=8
run 28=1,24=4
run 28=0,24=4
run 28=0,24=8
