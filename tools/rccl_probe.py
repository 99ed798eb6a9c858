"""RCCL over xGMI: reduce-scatter + all-gather of the few-shot step's gradient / parameter buffer (111.68 M fp32 = 447 MB) at N ranks, whole
buffer and per data-parallel bucket (embeddings | one per encoder layer | head), against the link bounds of DESIGN.md section 8
(7 links x 153 GB/s per GPU: RS + AG of 447 MB = 0.73 ms; a ring all-reduce bound by one link: 5.1 ms).  One process per GPU:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/rccl_probe.py [--iters 20] [--wire bf16]

At N = 1 it runs the same collectives on one rank (no wire traffic: launch + copy overhead only).  Prints ONE JSON line on rank 0."""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--wire", default="fp32", choices=["fp32", "bf16"])
    a = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    ones = torch.ones(1, device=dev)
    dist.all_reduce(ones)
    from cpt_amd import config as cfgmod
    from cpt_amd.engine import bucket_of
    from cpt_amd import synth
    cfg = cfgmod.oscar_base()
    # bucket sizes of the Oscar-base REC_MLM_CPT parameter layout, padded to multiples of 512 elements as engine.PackedModel does
    sizes = {}
    for name, shape, kind in synth.param_specs(cfg, "cpt"):
        if kind == "tied":
            continue
        n = 1
        for s in shape:
            n *= s
        k = bucket_of(name, cfg.num_hidden_layers)
        sizes[k] = sizes.get(k, 0) + n
    sizes = {k: (v + 511) // 512 * 512 for k, v in sizes.items()}
    total = sum(sizes.values())
    wdt = torch.bfloat16 if a.wire == "bf16" else torch.float32
    grad = torch.randn(total, device=dev, dtype=wdt)
    shard = torch.empty(total // world, device=dev, dtype=wdt)
    flat = torch.randn(total, device=dev)
    pshard = torch.empty(total // world, device=dev)

    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(a.iters):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.iters
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
    res = {"ranks": int(ones.item()), "elements": total, "wire": a.wire}
    res["reduce_scatter_whole_ms"] = round(timed(lambda: dist.reduce_scatter_tensor(shard, grad)) * 1e3, 4)
    res["all_gather_whole_ms"] = round(timed(lambda: dist.all_gather_into_tensor(flat, pshard)) * 1e3, 4)
    res["all_reduce_whole_ms"] = round(timed(lambda: dist.all_reduce(grad)) * 1e3, 4)
    offs, o = {}, 0
    for k in sorted(sizes):
        offs[k] = o
        o += sizes[k]

    def bucketed_rs():
        so = 0
        for k in sorted(sizes, reverse=True):          # backward order: head first
            n = sizes[k]
            dist.reduce_scatter_tensor(shard[so:so + n // world], grad[offs[k]:offs[k] + n])
            so += n // world

    def bucketed_ag():
        so = 0
        for k in sorted(sizes):
            n = sizes[k]
            dist.all_gather_into_tensor(flat[offs[k]:offs[k] + n], pshard[so:so + n // world])
            so += n // world
    res["reduce_scatter_bucketed_ms"] = round(timed(bucketed_rs) * 1e3, 4)
    res["all_gather_bucketed_ms"] = round(timed(bucketed_ag) * 1e3, 4)
    res["buckets"] = {str(k): sizes[k] for k in sorted(sizes)}
    wb = 2 if a.wire == "bf16" else 4
    if world > 1:
        res["bound_ms"] = {"reduce_scatter_7_links": round((world - 1) / world * total * wb / (7 * 153e9) * 1e3, 3),
                           "all_gather_7_links": round((world - 1) / world * total * 4 / (7 * 153e9) * 1e3, 3),
                           "ring_all_reduce_one_link": round(2 * (world - 1) / world * total * wb / 153e9 * 1e3, 3)}
        res["bus_GBs"] = {"reduce_scatter": round((world - 1) / world * total * wb / (res["reduce_scatter_whole_ms"] * 1e-3) / 1e9, 1),
                          "all_gather": round((world - 1) / world * total * 4 / (res["all_gather_whole_ms"] * 1e-3) / 1e9, 1)}
    if rank == 0:
        print(json.dumps(res), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
