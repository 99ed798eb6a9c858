#!/bin/bash
# Same-box A/B of the training step under cpt_set_tuning keys: tools/train_ab.sh "17=8" "17=16" ...  (first line: defaults)
O=gpurun_out
for B in 32 4; do
  echo "B=$B default: $(python bench.py --steps 20 --warmup 5 --mode train --batch $B --no-cpu 2>/dev/null | python -c 'import sys,json; print(json.loads(sys.stdin.readline())["ms_per_step"])')"
  for t in "$@"; do
    echo "B=$B $t: $(python bench.py --steps 20 --warmup 5 --mode train --batch $B --no-cpu --tune $t 2>/dev/null | python -c 'import sys,json; print(json.loads(sys.stdin.readline())["ms_per_step"])')"
  done
done
