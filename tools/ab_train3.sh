#!/bin/bash
# Same-box comparison of several cpt_set_tuning settings on the training step (development library).   usage: tools/ab_train3.sh <tag> "<k=v>" ...
T=$1; shift; O=gpurun_out; mkdir -p $O; : > $O/${T}_ab.txt
for rep in 1 2; do
  for BS in 32 4; do
    for V in "$@"; do
      ms=$(python bench.py --steps 30 --warmup 5 --mode train --batch $BS --no-cpu --no-sustained --tune "$V" 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
      echo "batch $BS tune $V: $ms ms" | tee -a $O/${T}_ab.txt
    done
  done
done
