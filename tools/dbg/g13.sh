run() { python bench.py --no-cpu --no-extra --no-sustained --workload $1 --steps 8 --warmup 3 --tune $2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernel_ms_per_step']
print('%-4s %-8s %.1f pairs/s ms/step %.3f ' % ('$1', '$2', d['value'], d['ms_per_step']), {a: round(b,3) for a,b in k.items()})"; }
run gqa 30=0; run gqa 30=96; run gqa 30=64; run gqa 30=128; run gqa 30=48; run gqa 30=0; run gqa 30=96
run vcr 30=0; run vcr 30=96; run vcr 30=32
python -m pytest tests/test_gpu_fullsize.py -q -x -k "config4 or config5" 2>&1 | tail -3
