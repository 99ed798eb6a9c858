import sys, time, torch
sys.path.insert(0, ".")
from cpt_amd import config as cfgmod, synth
from cpt_amd.modeling_rec import REC_MLM_CPT
dev = torch.device("cuda:0")
cfg = cfgmod.oscar_base()
m = REC_MLM_CPT(cfg); m.load_state_dict(synth.init_state_dict(cfg, 88, head="cpt")); m.tie_weights(); m.to(dev).eval().set_compute_dtype("bf16")
def mk(B, Lt, Li): return {k: v.to(dev) for k, v in synth.make_batch(B, cfg, seed=88, max_seq_len=Lt, img_seq_len=Li).items()}
def run(b):
    with torch.no_grad():
        return m(b["input_ids"], b["segment_ids"], b["attention_mask"], img_feats=b["img_feats"], mask_token_pos=b["mask_token_pos"])[0]
b64 = mk(64, 70, 50); bg = mk(256, 165, 45)
for _ in range(10): run(b64)
torch.cuda.synchronize()
for rep in range(3):
    ts = []
    for i in range(8):
        torch.cuda.synchronize(); t0 = time.perf_counter(); run(bg); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print("gqa per-step ms:", ["%.2f" % t for t in ts])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(5): run(bg)
    torch.cuda.synchronize(); print("5 back-to-back: %.2f ms/step" % ((time.perf_counter() - t0) * 1e3 / 5))
    for _ in range(20): run(b64)
