#!/bin/bash
# Final checks of the round on one box: the whole GPU suite, smoke, the default bench line (driver's flags), the training lines.   usage: tools/final_round.sh r06
T=$1; O=gpurun_out; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/${T}_gputests_product.log 2>&1; tail -4 $O/${T}_gputests_product.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/${T}_smoke.log 2>&1; tail -2 $O/${T}_smoke.log
python bench.py --steps 20 --warmup 5 > $O/${T}_bench_default_final.json 2> $O/${T}_bench_default_final.err; tail -c 300 $O/${T}_bench_default_final.json
python bench.py --steps 20 --warmup 5 --mode train --no-cpu > $O/${T}_bench_train_b32_final.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --mode train --batch 4 --no-cpu > $O/${T}_bench_train_b4_final.json 2>/dev/null
