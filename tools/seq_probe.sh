R=$PWD; O=$R/gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_dropout.py -x -q -m gpu 2>&1 | tail -3
for BS in 32 4; do python bench.py --steps 30 --warmup 5 --mode train --batch $BS --no-cpu --no-sustained 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B=$BS', d['ms_per_step'])"; done
cd /tmp && export TMPDIR=/tmp
rm -rf $O/seq_prof
rocprofv3 --kernel-trace --output-format csv -d $O/seq_prof -- python $R/bench.py --steps 4 --warmup 2 --mode train --batch 32 --no-cpu --no-sustained > /dev/null 2> $O/seq_prof.log
f=$(find $O/seq_prof -name "*kernel_trace.csv" | head -1)
python $R/tools/step_sequence.py $f > $O/r06_train_b32_step_sequence.txt
rm -rf $O/seq_prof
tail -5 $O/r06_train_b32_step_sequence.txt
