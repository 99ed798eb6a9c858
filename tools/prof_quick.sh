R=$PWD; O=$R/gpurun_out; cd /tmp; export TMPDIR=/tmp
rm -rf $O/tail_prof
rocprofv3 --kernel-trace --stats --output-format csv -d $O/tail_prof -- python $R/bench.py --steps 20 --warmup 5 --no-cpu --no-roofline --no-extra --no-io --no-live-pmc --no-sustained > $O/tail_bench.json 2> $O/tail_prof.log
cd $R
find $O/tail_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/tail_kernel_stats.csv
rm -rf $O/tail_prof
head -c 300 $O/tail_bench.json
