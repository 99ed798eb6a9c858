"""Phase stamps of the (sequence, three heads) QKV + attention kernel inside the bench forward (GPU box only).
usage: python tools/q3_timeline.py [abl]   -> per-workgroup mean ticks: prologue | K loop | Q/K/V tiles -> LDS | attention"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpt_amd import _lib as L, config as cfgmod, synth  # noqa: E402
from cpt_amd.modeling_rec import REC_MLM_CPT  # noqa: E402

dev = torch.device("cuda:0")
abl = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cfg = cfgmod.oscar_base()
model = REC_MLM_CPT(cfg)
model.load_state_dict(synth.init_state_dict(cfg, 88, head="cpt", randomize_all=False))
model.tie_weights()
model.to(dev).eval().set_compute_dtype("bf16")
bt = {k: v.to(dev) for k, v in synth.make_batch(64, cfg, seed=88).items()}


def fwd():
    with torch.no_grad():
        model(bt["input_ids"], bt["segment_ids"], bt["attention_mask"], img_feats=bt["img_feats"], mask_token_pos=bt["mask_token_pos"])


for _ in range(3):
    fwd()
buf = torch.zeros(4096 * 8, device=dev, dtype=torch.int64)
L.check(L.lib().cpt_set_tuning(1, abl))
L.check(L.lib().cpt_set_tuning(8, 10))     # the GEMMs share the trace buffer: stamp only the fused QKV + attention launches (epilogue id 10)
L.lib().cpt_debug_gemm_trace(buf.data_ptr())
fwd()
torch.cuda.synchronize()
L.lib().cpt_debug_gemm_trace(None)
L.check(L.lib().cpt_set_tuning(-1, 0))
t = buf.cpu().numpy().reshape(-1, 8)
t = t[t[:, 0] != 0]
us = (t[:, 6] - t[:, 5]) / 100.0                         # per-workgroup duration by the 100 MHz wall counter
span_us = (t[:, 6].max() - t[:, 5].min()) / 100.0
print("abl %d: %d workgroups; first start -> last end %.2f us; starts spread over %.2f us, ends over %.2f us; workgroup duration mean %.2f us (max %.2f); "
      "shader clock %.2f GHz" % (abl, len(t), span_us, (t[:, 5].max() - t[:, 5].min()) / 100.0, (t[:, 6].max() - t[:, 6].min()) / 100.0, us.mean(), us.max(),
                                 ((t[:, 4] - t[:, 0]) / us).mean() / 1e3))
print("  per-workgroup mean ticks: prologue %d | K loop %d | tiles -> LDS %d | attention %d | total %d" %
      ((t[:, 1] - t[:, 0]).mean(), (t[:, 2] - t[:, 1]).mean(), (t[:, 3] - t[:, 2]).mean(), (t[:, 4] - t[:, 3]).mean(), (t[:, 4] - t[:, 0]).mean()))
