"""CPU, world_size 2 over gloo: the bucketed reduce-scatter / sharded AdamW / all-gather machinery of the data-parallel
training step (cpt_amd.dist.ShardedGradSync, what train.FusedAdamW drives on the GPU) against a single-process AdamW step
on the mean gradient; bench.py's self-spawn command line.  Reference being replaced: DistributedDataParallel at
/root/reference/Oscar/oscar/fewshot/refcoco_cpt.py:516-522 and the result gather of utils/comm.py:102-142."""
import os
import socket
import subprocess
import sys
import tempfile

import pytest
import torch
import torch.multiprocessing as mp

from cpt_amd import dist as cdist
from oracle import cpt_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


BUCKETS = {0: (0, 1024), 1: (1024, 1536), 2: (1536, 3072), 3: (3072, 3584)}     # every range a multiple of 512
N = 3584
HP = dict(lr=1e-3, b1=0.9, b2=0.98, eps=1e-8, wd=0.01)


def _problem(world):
    g = torch.Generator().manual_seed(7)
    p0 = torch.randn(N, generator=g)
    grads = [[torch.randn(N, generator=g) for _ in range(world)] for _ in range(3)]      # 3 steps
    return p0, grads


def _sync_worker(rank, world, port, tmp, wire):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        p0, grads = _problem(world)
        flat = p0.clone() if rank == 0 else torch.zeros(N)
        cdist.broadcast_(flat, 0)                                   # replicas start from rank 0
        sync = cdist.ShardedGradSync(BUCKETS, "cpu", wire=wire)
        m = torch.zeros(sync.shard_elems)
        v = torch.zeros(sync.shard_elems)
        for step in range(3):
            grad = grads[step][rank].clone()
            sync.wait_params()                                       # forward would wait bucket by bucket
            sync.begin_backward()
            for k in (3, 2, 1):                                      # backward order: head, layers N-1 .. 0, embeddings
                sync.grads_ready(grad, k)
            sync.finish_reduce(grad)                                 # bucket 0 never reported: reduced here
            for k in sync.order:
                slo, shi = sync.shard_range(k)
                so = sync.soff[k]
                n = shi - slo
                gk = sync.gshard[so:so + n] / world                  # sum -> mean
                pn, mn, vn = O.adamw_step(flat[slo:shi], gk, m[so:so + n], v[so:so + n], step + 1, HP["lr"], HP["b1"], HP["b2"],
                                          HP["eps"], HP["wd"])
                flat[slo:shi] = pn
                m[so:so + n] = mn
                v[so:so + n] = vn
            sync.all_gather_params(flat)
        sync.wait_params()
        mfull = sync.gather_full(m, N)
        back = torch.zeros_like(m)
        sync.scatter_full(mfull, back)
        assert torch.equal(back, m)
        torch.save({"flat": flat, "m": mfull}, os.path.join(tmp, "r%d.pt" % rank))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("wire", [None, torch.bfloat16])
def test_sharded_grad_sync_matches_single_process_adamw(wire):
    world = 2
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_sync_worker, args=(world, _free_port(), tmp, wire), nprocs=world, join=True)
        r0 = torch.load(os.path.join(tmp, "r0.pt"))
        r1 = torch.load(os.path.join(tmp, "r1.pt"))
    assert torch.equal(r0["flat"], r1["flat"]) and torch.equal(r0["m"], r1["m"])        # replicas stay identical
    p, grads = _problem(world)
    m = torch.zeros(N)
    v = torch.zeros(N)
    for step in range(3):
        g = sum(grads[step]) / world
        p, m, v = O.adamw_step(p, g, m, v, step + 1, HP["lr"], HP["b1"], HP["b2"], HP["eps"], HP["wd"])
    tol = 1e-6 if wire is None else 2e-3
    assert (r0["flat"] - p).abs().max().item() < tol
    assert (r0["m"] - m).abs().max().item() < (1e-6 if wire is None else 1e-2)


def test_bucket_layout_is_contiguous_and_aligned():
    from cpt_amd import config as cfgmod
    from cpt_amd.engine import pack_order, bucket_of, BUCKET_ALIGN
    cfg = cfgmod.tiny()
    names = pack_order(cfg, "cpt")
    ks = [bucket_of(n, cfg.num_hidden_layers) for n in names]
    assert ks == sorted(ks)                                          # every bucket is one contiguous run
    assert ks[0] == 0 and ks[-1] == cfg.num_hidden_layers + 1
    assert BUCKET_ALIGN % (8 * 64) == 0                              # 1/2/4/8 ranks -> 64-element aligned shards
    assert bucket_of("bert.embeddings.word_embeddings.weight", 12) == 0      # tied table: completes with the lookup gradient
    assert bucket_of("bert.img_embedding.weight", 12) == 0
    assert bucket_of("bert.encoder.layer.11.output.dense.bias", 12) == 12
    assert bucket_of("cls.bias", 12) == 13 and bucket_of("bert.pooler.dense.weight", 12) == 13


def test_bench_spawns_one_rank_per_gpu():
    """`python bench.py --gpus N` without WORLD_SIZE re-executes itself under torch.distributed.run with N ranks."""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "3", "--warmup", "1",
                          "--print-launch"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert out.returncode == 0, out.stdout
    line = out.stdout.strip().splitlines()[-1]
    assert "torch.distributed.run" in line and "--nproc-per-node 4" in line and "--master-addr 127.0.0.1" in line
    assert line.rstrip().endswith("--gpus 4 --steps 3 --warmup 1")
