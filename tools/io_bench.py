"""Throughput of the region-feature wire-format decoder next to the reference's Python
path (oracle/io_oracle.py) on the same host cores, and -- with a GPU -- of the decode -> pinned staging -> H2D ->
forward pipeline against the forward alone.   python tools/io_bench.py [--seqs 64] [--threads N] [--gpu]"""
import argparse
import base64
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cpt_amd import io  # noqa: E402


def make_rows(n_rows, props, boxes, seed=0):
    rng = np.random.default_rng(seed)
    rows = []
    for r in range(n_rows):
        objs = [[{"rect": [1.0, 2.0, 30.0, 40.0], "bbox_id": j, "class": "dog", "conf": 0.9,
                  "feature": base64.b64encode(np.maximum(rng.standard_normal(2054), 0).astype(np.float32).tobytes()).decode()}
                 for j in range(boxes)] for _ in range(props)]
        rows.append(json.dumps({"objects": [objs, "a dog on the left", [["red"]] * props, [[[1, 2, 30, 40]]] * props]}).encode())
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seqs", type=int, default=64, help="sequences per batch (rows x proposals)")
    ap.add_argument("--props", type=int, default=8, help="proposal sequences per TSV row")
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--gpu", action="store_true")
    ap.add_argument("--depth", type=int, default=3)
    ap.add_argument("--dthreads", type=int, default=0)
    ap.add_argument("--workers", type=int, default=2)
    a = ap.parse_args()
    sys.path.insert(0, ROOT)
    from bench import usable_cores
    cores = a.threads or min(usable_cores(), 32)
    n_rows = a.seqs // a.props
    rows = make_rows(n_rows, a.props, 50)
    mb = sum(len(r) for r in rows) / 1e6
    out = torch.empty((a.seqs, 50, 2054)).pin_memory() if torch.cuda.is_available() else torch.empty((a.seqs, 50, 2054))
    mask = torch.empty((a.seqs, 50), dtype=torch.int64)

    def decode_batch(threads):
        for i, r in enumerate(rows):
            io.decode_row(r, out=out[i * a.props:(i + 1) * a.props], mask=mask[i * a.props:(i + 1) * a.props], threads=threads)

    def timeit(fn, n):
        fn()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        return (time.perf_counter() - t0) / n

    res = {"batch_seqs": a.seqs, "row_MB": round(mb / n_rows, 3), "cores": cores}
    t1 = timeit(lambda: decode_batch(1), 5)
    res["cpp_1_thread"] = {"seq_per_s": round(a.seqs / t1), "MB_per_s_text": round(mb / t1)}
    import concurrent.futures as cf
    nw = min(cores, n_rows)
    with cf.ThreadPoolExecutor(max_workers=nw) as ex:      # ctypes releases the GIL: rows decode concurrently

        def par():
            list(ex.map(lambda i: io.decode_row(rows[i], out=out[i * a.props:(i + 1) * a.props],
                                                mask=mask[i * a.props:(i + 1) * a.props], threads=1), range(n_rows)))
        tn = timeit(par, 5)
    res["cpp_%d_threads" % nw] = {"seq_per_s": round(a.seqs / tn), "MB_per_s_text": round(mb / tn)}
    for th in sorted(set([1, 4, 8, cores])):
        tb = timeit(lambda: io.decode_rows(rows, out=out, mask=mask, threads=th, max_seqs=a.seqs), 5)
        res["cpp_native_batch_%d_threads" % th] = {"seq_per_s": round(a.seqs / tb), "MB_per_s_text": round(mb / tb)}
    from oracle import io_oracle as IO

    def py_ref():
        for r in rows:
            feats = IO.decode_features(("x", r.decode()))[2]
            IO.pad_regions(feats, 50)
    tp = timeit(py_ref, 2)
    res["python_reference_1_thread"] = {"seq_per_s": round(a.seqs / tp), "MB_per_s_text": round(mb / tp)}
    res["speedup_1_thread"] = round(tp / t1, 1)

    if a.gpu and torch.cuda.is_available():
        from cpt_amd import config as cfgmod, synth
        from cpt_amd.modeling_rec import REC_MLM_CPT
        dev = torch.device("cuda:0")
        cfg = cfgmod.oscar_base()
        m = REC_MLM_CPT(cfg)
        m.load_state_dict(synth.init_state_dict(cfg, 88, head="cpt", randomize_all=False))
        m.tie_weights()
        m.to(dev).eval().set_compute_dtype("bf16")
        b = {k: v.to(dev) for k, v in synth.make_batch(a.seqs, cfg, seed=88).items()}

        def fwd(feats):
            with torch.no_grad():
                return m(b["input_ids"], b["segment_ids"], b["attention_mask"], img_feats=feats, mask_token_pos=b["mask_token_pos"])[0]
        for _ in range(5):
            fwd(b["img_feats"])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            fwd(b["img_feats"])
        torch.cuda.synchronize()
        tf = (time.perf_counter() - t0) / 20
        # pipeline: a decode thread pool fills pinned buffers, a side stream copies, the main stream runs the model
        import concurrent.futures as cf
        depth = a.depth
        host = [torch.empty((a.seqs, 50, 2054)).pin_memory() for _ in range(depth)]
        devb = [torch.empty((a.seqs, 50, 2054), device=dev) for _ in range(depth)]
        hmask = [torch.empty((a.seqs, 50), dtype=torch.int64) for _ in range(depth)]
        copy_stream = torch.cuda.Stream(dev)

        def decode_into(k, after=None):
            if after is not None:
                after.synchronize()               # the H2D copy out of this pinned buffer has finished
            io.decode_rows(rows, out=host[k], mask=hmask[k], threads=dthreads, max_seqs=a.seqs)
            return k

        dthreads = a.dthreads or max(1, min(cores - 2, 12))
        steps = 40
        sys.setswitchinterval(2e-4)          # the launch thread must not wait 5 ms for the interpreter lock
        with cf.ThreadPoolExecutor(max_workers=a.workers) as ex:
            # one C call per batch (native threads inside, GIL released), `depth` pinned buffers in rotation; the main
            # thread enqueues the H2D copy on the side stream and the forward on the compute stream
            futs = [ex.submit(decode_into, k) for k in range(depth)]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            t_wait = 0.0
            for s in range(steps):
                tw = time.perf_counter()
                k = futs[s].result()
                t_wait += time.perf_counter() - tw
                with torch.cuda.stream(copy_stream):
                    devb[k].copy_(host[k], non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(copy_stream)
                torch.cuda.current_stream().wait_event(ev)
                fwd(devb[k])
                futs.append(ex.submit(decode_into, k, ev))
            torch.cuda.synchronize()
            tpipe = (time.perf_counter() - t0) / steps
            t_wait /= steps
        res["gpu"] = {"forward_only_seq_per_s": round(a.seqs / tf), "decode_stage_forward_seq_per_s": round(a.seqs / tpipe),
                      "decode_threads": dthreads, "ms_per_step": round(tpipe * 1e3, 2),
                      "ms_waiting_for_decode": round(t_wait * 1e3, 2), "decode_alone_ms": round(timeit(lambda: decode_into(0), 5) * 1e3, 2), "h2d_MB_per_step": round(a.seqs * 50 * 2054 * 4 / 1e6, 1)}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
