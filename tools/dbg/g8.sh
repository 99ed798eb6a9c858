python -m pytest tests/test_gpu_model.py -q -x -k "x3" 2>&1 | tail -3
for t in 27=0 27=1 27=0 27=1; do python bench.py --no-cpu --no-extra --no-sustained --dtype bf16x3 --steps 10 --warmup 3 --tune $t 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernel_ms_per_step']
print('$t', d['value'], d['ms_per_step'], {a: round(b,3) for a,b in k.items()})"; done
python bench.py --no-extra --dtype bf16x3 --steps 10 --warmup 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['value'], d['parity'])"
