#!/bin/bash
# Same-box A/B of two BUILDS of the library on the training step: alternating runs at 32 and 4 sequences per GPU.
#   usage: tools/ab_lib.sh <tag> <libA.so> <libB.so>      -> gpurun_out/<tag>_ablib.txt
T=$1; A=$2; B=$3; O=gpurun_out; mkdir -p $O
: > $O/${T}_ablib.txt
for rep in 1 2; do
  for BS in 32 4; do
    for V in "$A" "$B"; do
      ms=$(CPT_LIB_PATH=$PWD/$V python bench.py --steps 30 --warmup 5 --mode train --batch $BS --no-cpu --no-sustained 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
      echo "batch $BS lib $V: $ms ms" | tee -a $O/${T}_ablib.txt
    done
  done
done
