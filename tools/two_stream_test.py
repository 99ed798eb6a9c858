import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpt_amd import config as cfgmod, synth, _lib, engine
from cpt_amd.modeling_rec import REC_MLM_CPT
dev = torch.device("cuda:0")
cfg = cfgmod.oscar_base()
model = REC_MLM_CPT(cfg)
model.load_state_dict(synth.init_state_dict(cfg, 88, head="cpt", randomize_all=False))
model.tie_weights()
model.to(dev).eval().set_compute_dtype("bf16")
B = 64
b = {k: v.to(dev) for k, v in synth.make_batch(B, cfg, seed=88).items()}
eng = model._engine()
def fwd(sl, wskey):
    eng._ws["fwd"] = eng._ws.get(wskey)
    out = eng.forward(b["input_ids"][sl], b["segment_ids"][sl], b["attention_mask"][sl], None, b["img_feats"][sl],
                      mask_pos=b["mask_token_pos"][sl], flags=_lib.OUT_MASK_LOGITS)
    eng._ws[wskey] = eng._ws["fwd"]
    return out["logits"]
def step1():
    with torch.no_grad():
        return fwd(slice(0, B), "a")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def step2(nsplit=2):
    outs = []
    cur = torch.cuda.current_stream()
    streams = [s1, s2]
    with torch.no_grad():
        for i in range(nsplit):
            st = streams[i % 2]
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                n = B // nsplit
                outs.append(fwd(slice(i * n, (i + 1) * n), "s%d" % i))
        for st in streams:
            cur.wait_stream(st)
    return outs
for name, fn in (("single stream B=64", step1), ("two streams 2x32", step2), ("single stream B=64", step1), ("two streams 2x32", step2)):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(40): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 40
    print("%-22s %.4f ms/step  %.0f pairs/s" % (name, dt * 1e3, B / dt))
a = step1(); bb = torch.cat(step2(), 0); torch.cuda.synchronize()
print("max diff", (a - bb).abs().max().item())
