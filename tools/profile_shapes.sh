#!/bin/bash
# rocprofv3 kernel tables of the secondary BASELINE shapes (configs[3] GQA B = 256, configs[4] Oscar-large VCR B = 32) and of the few-shot step at 4 sequences.
# usage (GPU box): tools/profile_shapes.sh r05   -> gpurun_out/<tag>_bench_{gqa_b256,vcr_large_b32}_bf16_kernel_stats.csv, <tag>_train_b4_kernel_stats.csv
T=$1; R=$PWD; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for w in gqa vcr; do
  rm -rf $O/${T}_prof_$w
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_prof_$w -- python $R/bench.py --steps 10 --warmup 3 --workload $w --no-cpu --no-roofline --no-extra > $O/${T}_bench_${w}_under_rocprof.json 2> $O/${T}_prof_$w.log
done
rm -rf $O/${T}_prof_b4
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_prof_b4 -- python $R/bench.py --steps 10 --warmup 3 --mode train --batch 4 --no-cpu > $O/${T}_bench_train_b4_under_rocprof.json 2> $O/${T}_prof_b4.log
cd $R
find $O/${T}_prof_gqa -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${T}_bench_gqa_b256_bf16_kernel_stats.csv
find $O/${T}_prof_vcr -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${T}_bench_vcr_large_b32_bf16_kernel_stats.csv
find $O/${T}_prof_b4 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${T}_train_b4_kernel_stats.csv
rm -rf $O/${T}_prof_gqa $O/${T}_prof_vcr $O/${T}_prof_b4
ls -la $O | grep -E "${T}_(bench_(gqa|vcr)|train_b4)"
