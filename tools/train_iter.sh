#!/bin/bash
# Training-step iteration on the GPU box: targeted parity tests, bench lines at 32 and 4 sequences per GPU, rocprofv3 kernel tables of both.
#   usage: tools/train_iter.sh <tag> [pytest -k expression]      -> gpurun_out/<tag>_*
T=$1; K=${2:-}; R=$PWD; O=$R/gpurun_out; mkdir -p $O
if [ -n "$K" ]; then
  timeout 1500 python -m pytest tests/test_gpu_dropout.py tests/test_gpu_bwd_ops.py tests/test_gpu_train.py -x -q -m gpu -k "$K" > $O/${T}_tests.log 2>&1
else
  timeout 1500 python -m pytest tests/test_gpu_dropout.py tests/test_gpu_bwd_ops.py tests/test_gpu_train.py -x -q -m gpu > $O/${T}_tests.log 2>&1
fi
tail -3 $O/${T}_tests.log
python bench.py --steps 20 --warmup 5 --mode train --no-cpu > $O/${T}_bench_train_b32_bf16.json 2> $O/${T}_bench_train_b32.err
python bench.py --steps 20 --warmup 5 --mode train --batch 4 --no-cpu > $O/${T}_bench_train_b4_bf16.json 2> $O/${T}_bench_train_b4.err
cd /tmp && export TMPDIR=/tmp
for B in 32 4; do
  rm -rf $O/${T}_prof_train$B
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_prof_train$B -- python $R/bench.py --steps 10 --warmup 3 --mode train --batch $B --no-cpu > $O/${T}_bench_train_b${B}_under_rocprof.json 2> $O/${T}_prof_train$B.log
  find $O/${T}_prof_train$B -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${T}_train_b${B}_kernel_stats.csv
  rm -rf $O/${T}_prof_train$B
done
cd $R
python - <<PY
import json
for b in (32, 4):
    try:
        d = json.loads(open("$O/${T}_bench_train_b%d_bf16.json" % b).read().strip().splitlines()[-1])
        print("B=%d: %.4f ms/step  sustained %s" % (b, d["ms_per_step"], d.get("sustained_2s", {}).get("ms_per_step")))
    except Exception as e:
        print("B=%d: no line (%s)" % (b, e))
PY
