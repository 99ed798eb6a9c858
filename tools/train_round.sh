#!/bin/bash
# Training-step A/B on the GPU box: bench lines at 32 and 4 sequences per GPU + rocprofv3 kernel stats of both.
# usage: tools/train_round.sh tag   -> gpurun_out/<tag>_train_*
T=$1; R=$PWD; O=$R/gpurun_out; TUNE=${2:+--tune $2}
python bench.py --steps 20 --warmup 5 --mode train --no-cpu $TUNE > $O/${T}_train_b32.json 2> $O/${T}_train_b32.err
python bench.py --steps 20 --warmup 5 --mode train --batch 4 --no-cpu $TUNE > $O/${T}_train_b4.json 2> $O/${T}_train_b4.err
cd /tmp && export TMPDIR=/tmp
for B in 32 4; do
  rm -rf $O/${T}_prof_train_b$B
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_prof_train_b$B -- python $R/bench.py --steps 10 --warmup 3 --mode train --batch $B --no-cpu $TUNE > /dev/null 2> $O/${T}_prof_train_b$B.log
  find $O/${T}_prof_train_b$B -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${T}_train_b${B}_kernel_stats.csv
  rm -rf $O/${T}_prof_train_b$B
done
cd $R
cat $O/${T}_train_b32.json $O/${T}_train_b4.json | cut -c1-400
