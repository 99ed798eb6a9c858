"""Does replaying the inference forward as a HIP graph shorten the step?  (one capture of model(...), replays against eager calls)"""
import sys, time, torch
sys.path.insert(0, ".")
from cpt_amd import config as cfgmod, synth
from cpt_amd.modeling_rec import REC_MLM_CPT
dev = torch.device("cuda:0")
cfg = cfgmod.oscar_base()
m = REC_MLM_CPT(cfg); m.load_state_dict(synth.init_state_dict(cfg, 88, head="cpt")); m.tie_weights(); m.to(dev).eval().set_compute_dtype("bf16")
b = {k: v.to(dev) for k, v in synth.make_batch(64, cfg, seed=88).items()}
def run():
    with torch.no_grad():
        return m(b["input_ids"], b["segment_ids"], b["attention_mask"], img_feats=b["img_feats"], mask_token_pos=b["mask_token_pos"])[0]
for _ in range(20): ref = run()
torch.cuda.synchronize()
def timeit(fn, n=200):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3 / n
print("eager   %.4f ms/step" % timeit(run))
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): run()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = run()
g.replay(); torch.cuda.synchronize()
print("graph output equal:", torch.equal(out, ref))
print("graph   %.4f ms/step" % timeit(g.replay))
print("eager   %.4f ms/step" % timeit(run))
print("graph   %.4f ms/step" % timeit(g.replay))
