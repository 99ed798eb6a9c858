python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29555 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-extra 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('torchrun n=1:', d['value'], d['n_gpus'], d['rccl_ranks'], d['sustained_2s'])"
python bench.py --mode train --steps 10 --warmup 3 --no-cpu --force-collectives 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d['comm']; print('train forced collectives:', d['value'], d['ms_per_step']); print({k: v for k, v in c.items() if k != 'per_bucket'}); print(list(c['per_bucket'].items())[:3])"
python bench.py --mode train --steps 10 --warmup 3 --no-cpu --force-collectives --grad-wire bf16 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); c = d['comm']; print('bf16 wire:', d['value'], c['reduce_scatter_ms_per_step'], c['all_gather_ms_per_step'], c['fraction_hidden'])"
python -c "
import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" 2>&1 | tail -3
