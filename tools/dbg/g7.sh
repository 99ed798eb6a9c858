run() { python bench.py --no-cpu --no-extra --no-sustained --steps 40 --warmup 10 --tune $1 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); k = d['kernel_ms_per_step']
print('%-18s ms/step %.4f  embed %.1f img %.1f head %.1f us' % ('$1', d['ms_per_step'], k['embed_ln']*1e3, k.get('img_proj',0)*1e3, k['head']*1e3))"; }
python -m pytest tests/test_gpu_model.py -q -x 2>&1 | tail -2
for r in 1 2; do
run 25=0,26=100
run 25=1,26=100
run 25=1,26=40
run 25=1,26=25
run 25=1,26=60
run 25=1,26=0
done
