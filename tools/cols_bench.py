"""Same-box timing of the inference forward with the full decoder against the decoder on the colour columns only (cpt_outputs.logit_cols).
usage: python tools/cols_bench.py [batch]   -> one JSON line"""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
from cpt_amd import config as cfgmod, synth          # noqa: E402
from cpt_amd.modeling_rec import REC_MLM_CPT         # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
cfg = cfgmod.oscar_base()
m = REC_MLM_CPT(cfg)
m.load_state_dict(synth.init_state_dict(cfg, 88, head="cpt", randomize_all=False))
m.tie_weights()
m.to(dev).eval().set_compute_dtype("bf16")
b = {k: v.to(dev) for k, v in synth.make_batch(B, cfg, seed=88, max_seq_len=70, img_seq_len=50).items()}
cols = torch.tensor(list(synth.COLOR_IDS) + [synth.NONE_ID], dtype=torch.int64, device=dev)


def run(vc, n):
    with torch.no_grad():
        for _ in range(n):
            out = m(b["input_ids"], b["segment_ids"], b["attention_mask"], img_feats=b["img_feats"], mask_token_pos=b["mask_token_pos"], vocab_columns=vc)[0]
    return out


res = {"batch": B, "columns": int(cols.numel())}
for rep in range(2):
    for name, vc in (("full", None), ("columns", cols)):
        run(vc, 20)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(vc, 400)
        torch.cuda.synchronize()
        res.setdefault(name + "_ms", []).append(round((time.perf_counter() - t0) / 400 * 1e3, 4))
print(json.dumps(res))
