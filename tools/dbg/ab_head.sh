for i in 1 2; do for l in cpt_amd/libcpt_hip.so tools/dbg/libcpt_dec192.so; do
  CPT_LIB_PATH=$l python bench.py --no-cpu --no-extra --no-sustained --steps 30 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['kernel_ms_per_step']
print('%-36s ms/step %.4f  head %.1f us img %.1f' % ('$l', d['ms_per_step'], k['head']*1e3, k['img_proj']*1e3))"
done; done
