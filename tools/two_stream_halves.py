"""Experiment (GPU box): the bench batch as ONE forward of B = 64 against TWO forwards of B = 32 on two streams (shared weights, one
workspace per stream).  python tools/two_stream_halves.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpt_amd import _lib as L, config as cfgmod, synth  # noqa: E402
from cpt_amd.modeling_rec import REC_MLM_CPT  # noqa: E402

dev = torch.device("cuda:0")
cfg = cfgmod.oscar_base()
m = REC_MLM_CPT(cfg)
m.load_state_dict(synth.init_state_dict(cfg, 88, head="cpt"))
m.tie_weights()
m.to(dev).eval().set_compute_dtype("bf16")
full = {k: v.to(dev) for k, v in synth.make_batch(64, cfg, seed=88).items()}
halves = [{k: v[i * 32:(i + 1) * 32].contiguous() for k, v in full.items()} for i in range(2)]
eng = m._engine()
streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
ws_by_stream = {}
orig_ws = eng.workspace


def ws_per_stream(B, Lt, Li, flags):
    key = torch.cuda.current_stream().cuda_stream
    need = orig_ws(B, Lt, Li, flags).numel()
    if key not in ws_by_stream or ws_by_stream[key].numel() < need:
        ws_by_stream[key] = torch.empty(need, device=dev, dtype=torch.uint8)
    return ws_by_stream[key]


def fwd(b):
    with torch.no_grad():
        return m(b["input_ids"], b["segment_ids"], b["attention_mask"], img_feats=b["img_feats"], mask_token_pos=b["mask_token_pos"])[0]


def timed(fn, steps=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


ref = fwd(full).clone()
print("one forward of 64:                     %.4f ms" % timed(lambda: fwd(full)))
L.check(L.lib().cpt_set_tuning(16, 96), "cpt_set_tuning")          # two-pass FFN-up (and with it panel mode) from 96 tiles on
print("two forwards of 32, one stream:        %.4f ms" % timed(lambda: [fwd(h) for h in halves]))
eng.workspace = ws_per_stream


def two():
    outs = []
    for s, h in zip(streams, halves):
        with torch.cuda.stream(s):
            outs.append(fwd(h))
    return outs


o = two()
torch.cuda.synchronize()
print("halves equal the full batch bit for bit:", torch.equal(torch.cat(o), ref))
print("two forwards of 32, two streams:       %.4f ms" % timed(two))
L.check(L.lib().cpt_set_tuning(-1, 0), "cpt_set_tuning")
