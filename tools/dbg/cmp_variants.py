"""debug: bf16 forward with GEMM variant A vs B (max |d logits|) at a few batch sizes"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cpt_amd import config as cfgmod, synth, _lib as L
from cpt_amd.modeling_rec import REC_MLM_CPT
dev = torch.device("cuda:0")
cfg = cfgmod.oscar_base()
m = REC_MLM_CPT(cfg)
m.load_state_dict(synth.init_state_dict(cfg, 88, head="cpt", randomize_all=False))
m.tie_weights()
m.to(dev).eval().set_compute_dtype("bf16")
va, vb = int(sys.argv[1]), int(sys.argv[2])
for B in (4, 7, 64):
    b = {k: v.to(dev) for k, v in synth.make_batch(B, cfg, seed=5).items()}
    outs = []
    for v in (va, vb, va):
        L.check(L.lib().cpt_set_tuning(0, v))
        with torch.no_grad():
            outs.append(m(b["input_ids"], b["segment_ids"], b["attention_mask"], img_feats=b["img_feats"], mask_token_pos=b["mask_token_pos"])[0].float().clone())
    L.check(L.lib().cpt_set_tuning(0, 3))
    d = (outs[0] - outs[1]).abs()
    print("B=%d variant %d vs %d: max|d| %.3e (rows worst: %s)  repeat of %d bit-equal: %s  finite %s" % (B, va, vb, d.max().item(), d.max(1).values.topk(min(3, B)).indices.tolist(), va, torch.equal(outs[0], outs[2]), torch.isfinite(outs[0]).all().item()))
