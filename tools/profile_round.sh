#!/bin/bash
# Round artefacts on the GPU box: PMC passes first (bench.py's roofline.traffic reads their summary), bench lines, rocprofv3
# kernel stats of the bench command.   usage: tools/profile_round.sh r02   -> gpurun_out/<tag>_*
T=$1; R=$PWD; O=$R/gpurun_out
tools/pmc_bench.sh gpurun_out/${T}_pmc > $O/${T}_pmc.log 2>&1
cp $O/${T}_pmc/summary.json $O/${T}_pmc_bench_summary.json 2>/dev/null
cp $O/${T}_pmc/summary.json $R/profiles/${T}_pmc_bench_summary.json 2>/dev/null      # box-local: read by bench.py below
python bench.py --steps 20 --warmup 5 > $O/${T}_bench_b64_bf16.json 2> $O/${T}_bench_b64_bf16.err
python bench.py --steps 10 --warmup 3 --dtype fp32 --no-cpu > $O/${T}_bench_b64_fp32.json 2>/dev/null
python bench.py --steps 10 --warmup 3 --dtype bf16x3 --no-cpu > $O/${T}_bench_b64_bf16x3.json 2>/dev/null
python bench.py --steps 10 --warmup 3 --workload gqa > $O/${T}_bench_gqa_b256_bf16.json 2>/dev/null
python bench.py --steps 10 --warmup 3 --workload vcr > $O/${T}_bench_vcr_large_b32_bf16.json 2>/dev/null
python bench.py --steps 10 --warmup 3 --mode train > $O/${T}_bench_train_b32_bf16.json 2>/dev/null
python bench.py --steps 5 --warmup 2 --mode train --dtype bf16x3 --no-cpu > $O/${T}_bench_train_b32_bf16x3.json 2>/dev/null
python bench.py --steps 3 --warmup 1 --mode train --dtype fp32 --no-cpu > $O/${T}_bench_train_b32_fp32.json 2>/dev/null
python bench.py --steps 5 --warmup 2 --mode train --workload gqa > $O/${T}_bench_train_gqa_b32_bf16.json 2>/dev/null
python bench.py --steps 5 --warmup 2 --mode train --workload vcr > $O/${T}_bench_train_vcr_large_b8_bf16.json 2>/dev/null
python bench.py --steps 10 --warmup 3 --all-rows --no-cpu > $O/${T}_bench_b64_bf16_allrows.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rm -rf $O/${T}_prof; rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_prof -- python $R/bench.py --steps 20 --warmup 5 --no-cpu --no-roofline --no-extra > $O/${T}_bench_under_rocprof.json 2> $O/${T}_prof.log
rm -rf $O/${T}_prof_train; rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_prof_train -- python $R/bench.py --steps 10 --warmup 3 --mode train --no-cpu > $O/${T}_bench_train_under_rocprof.json 2> $O/${T}_prof_train.log
cd $R
find $O/${T}_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${T}_bench_b64_bf16_kernel_stats.csv
find $O/${T}_prof_train -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${T}_bench_train_b32_bf16_kernel_stats.csv
# keep the merge-back small
rm -rf $O/${T}_prof $O/${T}_prof_train
find $O/${T}_pmc -name "*.csv" -size +1M -delete 2>/dev/null
CPT_AMD_ABLATION=1 python tools/panel_model_trace.py > $O/${T}_panel_model_trace.txt 2>&1
CPT_AMD_ABLATION=1 python tools/q3_timeline.py > $O/${T}_q3_timeline.txt 2>&1
CPT_AMD_ABLATION=1 python tools/panel_bench.py --rounds 2 > $O/${T}_panel_bench.txt 2>&1
python tools/rp_bench.py > $O/${T}_rp_bench.txt 2>&1
ls -la $O | grep ${T}_
