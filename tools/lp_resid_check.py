"""A/B of the bf16-residual option (cpt_set_tuning(4, 1)): accuracy vs the fp32 oracle and speed."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpt_amd import config as cfgmod, synth, _lib as L
from cpt_amd.modeling_rec import REC_MLM_CPT
from oracle import cpt_oracle as O
dev = torch.device("cuda:0")
cfg = cfgmod.oscar_base()
sd = synth.init_state_dict(cfg, 88, head="cpt")
m = REC_MLM_CPT(cfg); m.load_state_dict(sd); m.tie_weights(); m.to(dev).eval().set_compute_dtype("bf16")
b = synth.make_batch(8, cfg, seed=21)
with torch.no_grad():
    ref = O.rec_mlm_cpt_forward(sd, cfg.to_dict(), b["input_ids"], b["segment_ids"], b["attention_mask"], img_feats=b["img_feats"], mask_rows_only=b["mask_token_pos"])[0]
d = {k: v.to(dev) for k, v in b.items()}
big = {k: v.to(dev) for k, v in synth.make_batch(64, cfg, seed=88).items()}
for flag in (0, 1, 0, 1):
    L.check(L.lib().cpt_set_tuning(4, flag))
    with torch.no_grad():
        got = m(d["input_ids"], d["segment_ids"], d["attention_mask"], img_feats=d["img_feats"], mask_token_pos=d["mask_token_pos"])[0].cpu()
        for _ in range(5): m(big["input_ids"], big["segment_ids"], big["attention_mask"], img_feats=big["img_feats"], mask_token_pos=big["mask_token_pos"])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(40): m(big["input_ids"], big["segment_ids"], big["attention_mask"], img_feats=big["img_feats"], mask_token_pos=big["mask_token_pos"])
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 40
    err = (got - ref).abs()
    cols = list(synth.COLOR_IDS)
    same = (got[:, cols].argmax(1) == ref[:, cols].argmax(1)).float().mean().item()
    print("bf16 residual=%d: max|dlogit| %.4f mean %.5f colour-argmax agreement %.2f   %.4f ms/step" % (flag, err.max(), err.mean(), same, dt * 1e3))
L.check(L.lib().cpt_set_tuning(4, 0))
