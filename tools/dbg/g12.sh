R=$PWD; O=$R/gpurun_out
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 tools/rccl_probe.py > $O/r04_rccl_probe_n1.json 2> $O/r04_rccl_probe_n1.err; tail -2 $O/r04_rccl_probe_n1.json | cut -c1-600
tools/yardstick.bin > $O/r04_yardstick.json 2> $O/r04_yardstick.err; head -8 $O/r04_yardstick.json
tools/pmc_rowkernels.sh r04_pmc_rows > $O/r04_pmc_rows.log 2>&1; tail -5 $O/r04_pmc_rows.log | cut -c1-300
python tools/io_pipeline_bench.py --workers 6,8 --threads 2 --out $O/r04_io_pipeline.json > $O/r04_io_pipeline.log 2>&1; tail -4 $O/r04_io_pipeline.log | cut -c1-400
python bench.py --steps 20 --warmup 5 --mode train --no-cpu > $O/r04_train_b32_comm_probe.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/r04_train_b32_comm_probe.json')); print('train', d['value'], d['ms_per_step'], d.get('comm'), d['sustained_2s'])"
