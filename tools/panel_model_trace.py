"""Phase stamps of the panel LayerNorm producers (gemm_prod.hip) INSIDE the bench forward (GPU box only):
python tools/panel_model_trace.py [--batch 64]  -- prints, for attn-out (K = 768) and FFN-down (K = 3072) of the last layer, the mean
ticks per workgroup of prologue / K loop / epilogue, the shader clock and the wall time from first start to last end."""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpt_amd import _lib as L, config as cfgmod, synth  # noqa: E402
from cpt_amd.modeling_rec import REC_MLM_CPT  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    cfg = cfgmod.oscar_base()
    m = REC_MLM_CPT(cfg)
    m.load_state_dict(synth.init_state_dict(cfg, 88, head="cpt"))
    m.tie_weights()
    m.to(dev).eval().set_compute_dtype("bf16")
    d = {k: v.to(dev) for k, v in synth.make_batch(a.batch, cfg, seed=88).items()}
    lib = L.lib()

    def fwd():
        with torch.no_grad():
            return m(d["input_ids"], d["segment_ids"], d["attention_mask"], img_feats=d["img_feats"], mask_token_pos=d["mask_token_pos"])[0]
    for _ in range(5):
        fwd()
    M = a.batch * 120
    nwg = (M // 128) * 4
    for name, K in (("attn_out", 768), ("ffn_down", 3072)):
        tr = torch.zeros(nwg * 8, dtype=torch.int64, device=dev)
        L.check(lib.cpt_set_tuning(8, 11 | (K << 8)), "cpt_set_tuning")
        lib.cpt_debug_gemm_trace(C.c_void_p(tr.data_ptr()))
        for _ in range(3):
            fwd()
        torch.cuda.synchronize()
        lib.cpt_debug_gemm_trace(None)
        t = tr.view(nwg, 8).cpu()
        dur = (t[:, 4] - t[:, 0]).float()
        wdur = (t[:, 5] - t[:, 3]).float() * 0.01
        pro, kl, ep = (t[:, 1] - t[:, 0]).float().mean().item(), (t[:, 2] - t[:, 1]).float().mean().item(), (t[:, 4] - t[:, 2]).float().mean().item()
        print("%-9s in the model (last layer): mean ticks per workgroup: prologue %.0f  K loop %.0f (%.0f per K-tile)  epilogue %.0f; shader clock %.2f GHz; "
              "workgroup wall time mean %.2f max %.2f us; first start -> last end %.2f us"
              % (name, pro, kl, kl / (K // 64), ep, (dur / wdur).mean().item() * 1e-3, wdur.mean().item(), wdur.max().item(),
                 (t[:, 5].max() - t[:, 3].min()).item() * 0.01), flush=True)
    L.check(lib.cpt_set_tuning(-1, 0), "cpt_set_tuning")


if __name__ == "__main__":
    main()
