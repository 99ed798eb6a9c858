"""debug: GQA-shape batch invariance (B=256 rows 0..3 vs B=4) per GEMM variant"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cpt_amd import config as cfgmod, synth, _lib as L
from cpt_amd.modeling_rec import REC_MLM_CPT
dev = torch.device("cuda:0")
cfg = cfgmod.oscar_base()
m = REC_MLM_CPT(cfg)
m.load_state_dict(synth.init_state_dict(cfg, 88, head="cpt"))
m.tie_weights()
m.to(dev).eval().set_compute_dtype("bf16")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
b = synth.make_batch(B, cfg, seed=41, max_seq_len=165, img_seq_len=45, vary_regions=True)
d = {k: v.to(dev) for k, v in b.items()}
ds = {k: v[:4].contiguous() for k, v in d.items()}
for v in (3, 15, 20):
    L.check(L.lib().cpt_set_tuning(0, v))
    with torch.no_grad():
        big = m(d["input_ids"], d["segment_ids"], d["attention_mask"], img_feats=d["img_feats"], mask_token_pos=d["mask_token_pos"])[0]
        small = m(ds["input_ids"], ds["segment_ids"], ds["attention_mask"], img_feats=ds["img_feats"], mask_token_pos=ds["mask_token_pos"])[0]
    L.check(L.lib().cpt_set_tuning(0, 3))
    dd = (big[:4] - small).abs().max(1).values
    print("variant %d: per-sequence max|d| %s" % (v, ["%.2e" % t for t in dd.tolist()]))
for fold, fuse in ((0, 1), (1, 0)):
    L.check(L.lib().cpt_set_tuning(5, fold)); L.check(L.lib().cpt_set_tuning(6, fuse))
    with torch.no_grad():
        big = m(d["input_ids"], d["segment_ids"], d["attention_mask"], img_feats=d["img_feats"], mask_token_pos=d["mask_token_pos"])[0]
        small = m(ds["input_ids"], ds["segment_ids"], ds["attention_mask"], img_feats=ds["img_feats"], mask_token_pos=ds["mask_token_pos"])[0]
    L.check(L.lib().cpt_set_tuning(5, 1)); L.check(L.lib().cpt_set_tuning(6, 1))
    dd = (big[:4] - small).abs().max(1).values
    print("fold %d fuse %d: per-sequence max|d| %s" % (fold, fuse, ["%.2e" % t for t in dd.tolist()]))
