import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from cpt_amd import config as cfgmod, synth
from cpt_amd.modeling_rec import REC_MLM_CPT
from cpt_amd.train import FusedAdamW
dev = torch.device("cuda:0")
cfg = cfgmod.oscar_base()
cfg.hidden_dropout_prob = cfg.attention_probs_dropout_prob = 0.0
m = REC_MLM_CPT(cfg)
m.load_state_dict(synth.init_state_dict(cfg, 88, head="cpt"))
m.tie_weights()
m.to(dev).train().set_compute_dtype("bf16")
lr = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-4
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
task = sys.argv[3] if len(sys.argv) > 3 else "copy"
opt = FusedAdamW(m, lr=lr, betas=(0.9, 0.98), weight_decay=0.01)
cols = torch.tensor(list(synth.COLOR_IDS))
def batch(seed, B=32):
    b = synth.make_batch(B, cfg, seed=seed, vary_regions=True)
    if task == "copy":          # label = the colour word that appears in the od-label text
        ids = b["input_ids"]
        isc = (ids[:, :, None] == cols[None, None, :]).any(-1)
        pos = isc.float().argmax(1)
        b["colors"] = ids[torch.arange(B), pos]
    if task == "paint":         # CPT-like: the regions are "painted": a colour-specific block of feature dims is raised in every region
        g = torch.Generator().manual_seed(seed)
        c = torch.randint(0, len(cols), (B,), generator=g)
        for i in range(B):
            b["img_feats"][i, :, 64 * int(c[i]): 64 * int(c[i]) + 64] += float(os.environ.get("PAINT", "4.0"))
        b["colors"] = cols[c]
    return {k: v.to(dev) for k, v in b.items()}
t0 = time.time()
for s in range(steps):
    b = batch(5000 + s)
    opt.zero_grad()
    loss, _ = m(b["input_ids"], b["segment_ids"], b["attention_mask"], img_feats=b["img_feats"], masked_lm_labels=b["colors"], mask_token_pos=b["mask_token_pos"])
    loss.backward()
    opt.step()
    if s % 50 == 0 or s == steps - 1:
        m.eval()
        with torch.no_grad():
            e = batch(900000 + s, 64)
            lg = m(e["input_ids"], e["segment_ids"], e["attention_mask"], img_feats=e["img_feats"], mask_token_pos=e["mask_token_pos"])[0].float().cpu()
        c = lg[:, cols]
        t2 = c.topk(2, 1).values
        acc = (cols[c.argmax(1)] == e["colors"].cpu()).float().mean().item()
        print("step %d loss %.4f eval: colour acc %.2f median margin %.3f min margin %.4f  (%.1fs)" % (s, loss.item(), acc, (t2[:,0]-t2[:,1]).median().item(), (t2[:,0]-t2[:,1]).min().item(), time.time()-t0), flush=True)
        m.train()
