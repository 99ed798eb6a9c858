"""Per-CU timeline of the pipelined GEMM's workgroups (CPT_ABLATION build: shader-clock stamps + HW_ID).  GPU box only.
usage: python tools/gemm_cu_timeline.py <variant> <shape> [skew]      (shape: qkv|attn_out|ffn_up|ffn_down)
Shows which workgroups share a CU and how their K-loop / epilogue phases overlap."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpt_amd import _lib as L, ops  # noqa: E402

dev = torch.device("cuda:0")
dt = torch.bfloat16
M, H, I = 64 * 120, 768, 3072
shapes = {"qkv": (M, 3 * H, H, L.EPI_NONE, dt), "attn_out": (M, H, H, L.EPI_RESID, torch.float32),
          "ffn_up": (M, I, H, L.EPI_GELU, dt), "ffn_down": (M, H, I, L.EPI_RESID, torch.float32)}
variant, name = int(sys.argv[1]), sys.argv[2]
skew = int(sys.argv[3]) if len(sys.argv) > 3 else 0
abl = int(sys.argv[4]) if len(sys.argv) > 4 else 0
buf = torch.zeros(4096 * 8, device=dev, dtype=torch.int64)
if name.startswith("model:"):
    # the kernels of the bf16 bench forward itself: model:<epilogue id>:<K>  (6 LN producer, 8 FFN-up, 10 QKV + attention)
    from cpt_amd import config as cfgmod, synth
    from cpt_amd.modeling_rec import REC_MLM_CPT
    _, epi_id, kf = name.split(":")
    cfg = cfgmod.oscar_base()
    model = REC_MLM_CPT(cfg)
    model.load_state_dict(synth.init_state_dict(cfg, 88, head="cpt", randomize_all=False))
    model.tie_weights()
    model.to(dev).eval().set_compute_dtype("bf16")
    bt = {k: v.to(dev) for k, v in synth.make_batch(64, cfg, seed=88).items()}
    L.check(L.lib().cpt_set_tuning(0, variant))
    L.check(L.lib().cpt_set_tuning(7, skew))

    def fwd():
        with torch.no_grad():
            model(bt["input_ids"], bt["segment_ids"], bt["attention_mask"], img_feats=bt["img_feats"], mask_token_pos=bt["mask_token_pos"])
    for _ in range(3):
        fwd()
    L.check(L.lib().cpt_set_tuning(8, int(epi_id) | (int(kf) << 8)))
    L.check(L.lib().cpt_set_tuning(1, abl))
    L.lib().cpt_debug_gemm_trace(buf.data_ptr())
    fwd()
    torch.cuda.synchronize()
    L.lib().cpt_debug_gemm_trace(None)
    L.check(L.lib().cpt_set_tuning(8, 255))
    L.check(L.lib().cpt_set_tuning(1, 0))
else:
    m, n, k, epi, odt = shapes[name]
    torch.manual_seed(0)
    x = torch.randn(m, k, device=dev).to(dt)
    w = (torch.randn(n, k, device=dev) * 0.05).to(dt)
    b = torch.randn(n, device=dev)
    r = torch.randn(m, n, device=dev) if epi == L.EPI_RESID else None
    L.check(L.lib().cpt_set_tuning(0, variant))
    L.check(L.lib().cpt_set_tuning(7, skew))
    for _ in range(3):
        ops.gemm(x, w, b, epi=epi, resid=r, out_dtype=odt)
    L.lib().cpt_debug_gemm_trace(buf.data_ptr())
    ops.gemm(x, w, b, epi=epi, resid=r, out_dtype=odt)
    torch.cuda.synchronize()
    L.lib().cpt_debug_gemm_trace(None)
L.check(L.lib().cpt_set_tuning(0, 3))
L.check(L.lib().cpt_set_tuning(7, 0))
t = buf.cpu().numpy().reshape(-1, 8)
print("debug counter (last trace slot):", int(t[-1, 7]))
t[-1, :] = 0
t = t[t[:, 0] != 0]
if os.environ.get("CPT_TRACE_SUMS"):
    # gemm_ffn.hip's two-pass kernel: t[3] = pass boundary, t[5..7] = wave 0's sums of (retire + vmcnt wait, barrier wait, DMA issue)
    print("abl %d 2-pass kernel, %d WGs, mean ticks: prologue %d  pass0 %d  pass1 %d  tail-epilogue %d | in the K loops: retire+vmcnt %d  barrier %d  dma-issue %d  rest %d"
          % (abl, len(t), (t[:, 1] - t[:, 0]).mean(), (t[:, 3] - t[:, 1]).mean(), (t[:, 2] - t[:, 3]).mean(), (t[:, 4] - t[:, 2]).mean(),
             t[:, 5].mean(), t[:, 6].mean(), t[:, 7].mean(), (t[:, 2] - t[:, 1] - t[:, 5] - t[:, 6] - t[:, 7]).mean()))
    sys.exit(0)
hw, xcc, bid = t[:, 5], t[:, 6] & 15, t[:, 7]
cu = ((hw >> 8) & 15) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5) | (xcc << 8)
print("abl %d " % abl, end="")
print("%s variant %d skew %d: %d workgroups on %d distinct CUs (xcc values %s)" % (name, variant, skew, len(t), len(set(cu.tolist())), sorted(set(xcc.tolist()))))
both_k = one_k = both_e = 0
shown = 0
for c in sorted(set(cu.tolist())):
    sel = np.where(cu == c)[0]
    sel = sel[np.argsort(t[sel, 0])]
    base = t[sel, 0].min()
    # event sweep over this CU's workgroups: state 1 = K loop [t1,t2), state 2 = epilogue [t2,t4)
    ev = []
    for i in sel:
        ev += [(t[i, 1] - base, 0, +1), (t[i, 2] - base, 0, -1), (t[i, 2] - base, 1, +1), (t[i, 4] - base, 1, -1)]
    ev.sort()
    nk = ne = 0
    last = 0
    for tm, kind, d in ev:
        dur = tm - last
        if nk >= 2: both_k += dur
        elif nk == 1: one_k += dur
        elif ne >= 1: both_e += dur
        last = tm
        if kind == 0: nk += d
        else: ne += d
    if shown < 4:
        shown += 1
        print(" CU %4x:" % c, "  ".join("wg%-4d[%6d k%6d e%6d end%6d]" % (bid[i], t[i, 0] - base, t[i, 1] - base, t[i, 2] - base, t[i, 4] - base) for i in sel))
tot = both_k + one_k + both_e
print(" CU-time split: >=2 WGs in K loop %.1f %%, exactly 1 in K loop %.1f %%, none in K loop (epilogue only) %.1f %%" % (100.0 * both_k / tot, 100.0 * one_k / tot, 100.0 * both_e / tot))
span = (t[:, 4].max() - t[:, 0].min())
print(" kernel span %d ticks; per-WG mean: prologue %d  k-loop %d  epilogue %d" % (span, (t[:, 1] - t[:, 0]).mean(), (t[:, 2] - t[:, 1]).mean(), (t[:, 4] - t[:, 2]).mean()))
