"""decode -> shared pinned ring -> H2D -> forward with worker PROCESSES (cpt_amd.io.DecodePool), against the forward alone
(GPU box).   python tools/io_pipeline_bench.py [--workers 6 --threads 2 --steps 60]"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from cpt_amd import config as cfgmod, io, synth  # noqa: E402
from io_bench import make_rows  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workers", default="6")
    ap.add_argument("--threads", default="2")
    ap.add_argument("--rows", type=int, default=32, help="rows of the synthetic predictions file (8 proposals x 50 boxes each)")
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    from cpt_amd.modeling_rec import REC_MLM_CPT
    dev = torch.device("cuda:0")
    tsv_path = "/tmp/cpt_io_pipeline.tsv"
    if not os.path.exists(tsv_path):
        rows = make_rows(a.rows, 8, 50)
        with open(tsv_path, "wb") as f:
            for i, r in enumerate(rows):
                f.write(b"img_%d\t" % i + r + b"\n")
        io.generate_lineidx_file(tsv_path, os.path.splitext(tsv_path)[0] + ".lineidx")
    n_rows = io.TSVFile(tsv_path).num_rows()
    cfg = cfgmod.oscar_base()
    m = REC_MLM_CPT(cfg)
    m.load_state_dict(synth.init_state_dict(cfg, 88, head="cpt", randomize_all=False))
    m.tie_weights()
    m.to(dev).eval().set_compute_dtype("bf16")
    B = 64
    b = {k: v.to(dev) for k, v in synth.make_batch(B, cfg, seed=1).items()}

    def fwd(feats, mask):
        with torch.no_grad():
            return m(b["input_ids"], b["segment_ids"], mask, img_feats=feats, mask_token_pos=b["mask_token_pos"])[0]
    for _ in range(5):
        fwd(b["img_feats"], b["attention_mask"])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        fwd(b["img_feats"], b["attention_mask"])
    torch.cuda.synchronize()
    fwd_only = B * a.steps / (time.perf_counter() - t0)
    res = {"forward_only_seq_per_s": round(fwd_only), "batch_seqs": B, "configs": []}
    print("forward only: %.0f seq/s" % fwd_only, flush=True)
    batches = [[(8 * i + j) % n_rows for j in range(8)] for i in range(a.steps + 16)]
    for workers, threads in [(int(w), int(t)) for w in a.workers.split(",") for t in a.threads.split(",")]:
        pool = io.DecodePool(tsv_path, max_seqs=B, workers=workers, slots=2 * workers, threads=threads)
        side = torch.cuda.Stream(dev)
        dfe = [torch.empty((B, 50, 2054), device=dev) for _ in range(2)]
        dma = [torch.empty((B, 120), dtype=torch.int64, device=dev) for _ in range(2)]
        for k in range(2):
            dma[k][:, :70] = b["attention_mask"][:, :70]
        consumed = [None, None]
        nxt = 0

        def pump():
            nonlocal nxt
            while pool.can_submit() and nxt < len(batches):
                pool.submit(batches[nxt])
                nxt += 1
        pump()
        t_wait = 0.0
        t0 = None
        for step in range(a.steps + 8):
            if step == 8:                                   # steady state from here
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                t_wait = 0.0
            tw = time.perf_counter()
            slot, names, infos, spr, regions = pool.next()
            t_wait += time.perf_counter() - tw
            S = sum(spr)
            k = step & 1
            with torch.cuda.stream(side):
                if consumed[k] is not None:
                    side.wait_event(consumed[k])            # the forward that read this device buffer is done
                dfe[k][:S].copy_(pool.feats[slot][:S], non_blocking=True)
                dma[k][:S, 70:].copy_(pool.masks[slot][:S], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(side)
            torch.cuda.current_stream().wait_event(ev)
            fwd(dfe[k][:S], dma[k][:S])
            consumed[k] = torch.cuda.Event()
            consumed[k].record()
            ev.synchronize()                                # the copy out of the pinned slot is done: the slot is free
            pool.release(slot)
            pump()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        rate = B * a.steps / dt
        pool.close()
        r = {"workers": workers, "threads_per_worker": threads, "seq_per_s": round(rate), "ms_per_step": round(dt / a.steps * 1e3, 3),
             "ms_waiting_for_decode_per_step": round(t_wait / a.steps * 1e3, 3), "fraction_of_forward_only": round(rate / fwd_only, 3)}
        res["configs"].append(r)
        print(json.dumps(r), flush=True)
    if a.out:
        json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
