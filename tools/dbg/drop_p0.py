"""debug: which tensor differs between two p=0 training steps (atomics?)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cpt_amd import config as cfgmod, synth
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from test_gpu_dropout import _model
dev = torch.device("cuda:0")
cfg = cfgmod.tiny()
b = {k: v.to(dev) for k, v in synth.make_batch(4, cfg, seed=9, max_seq_len=20, img_seq_len=6).items()}
def run(m):
    for prm in m.parameters():
        prm.grad = None
    loss, _ = m(b["input_ids"], b["segment_ids"], b["attention_mask"], img_feats=b["img_feats"], masked_lm_labels=b["colors"], mask_token_pos=b["mask_token_pos"])
    loss.backward()
    return {"loss": loss.detach().clone(), **{n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}}
for mode in ("fp32", "bf16"):
    m0 = _model(cfgmod.tiny(), dev, mode, 0.0)
    a = run(m0); a2 = run(m0)
    bad = [n for n in a if not torch.equal(a[n], a2[n])]
    print(mode, "train p=0 twice: differing:", bad[:8], len(bad))
    m0.eval()
    e = run(m0)
    bad = [(n, float((a[n] - e[n]).abs().max())) for n in a if not torch.equal(a[n], e[n])]
    print(mode, "train vs eval: differing:", bad[:8], len(bad))
    m0.config.hidden_dropout_prob = m0.config.attention_probs_dropout_prob = 0.1
    e2 = run(m0)
    bad = [(n, float((a[n] - e2[n]).abs().max())) for n in a if not torch.equal(a[n], e2[n])]
    print(mode, "train vs eval(p cfg 0.1): differing:", bad[:8], len(bad))
