"""debug: do rows of a big problem equal the same rows computed as a small problem, per operator?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from cpt_amd import _lib as L, ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
H, I = 768, 3072
for Mbig, Msmall in ((53760, 840), (7680, 960), (7680, 120)):
    x = torch.randn(Mbig, H, device=dev) * 1.2 + 0.3
    a = x.to(torch.bfloat16)
    st = ops.row_stats_table(x)
    for name, N, gelu in (("qkv (consumer, no gelu)", 3 * H, False), ("ffn-up (consumer, gelu)", I, True)):
        wf = (torch.randn(N, H, device=dev) * 0.03).to(torch.bfloat16)
        colc = wf.float().sum(1).contiguous(); cold = torch.randn(N, device=dev) * 0.1
        big = ops.gemm_ln_cons(a, wf, st, colc, cold, 1e-12, H, gelu)
        small = ops.gemm_ln_cons(a[:Msmall].contiguous(), wf, st[:Msmall].contiguous(), colc, cold, 1e-12, H, gelu)
        d = (big[:Msmall].float() - small.float()).abs()
        bad_rows = (d.max(1).values > 0).nonzero().flatten()
        print("M %d vs %d  %-26s equal %s  (differing rows: %d, first %s)" % (Mbig, Msmall, name, torch.equal(big[:Msmall], small), bad_rows.numel(), bad_rows[:4].tolist()))
    for name, K in (("attn-out (producer)", H), ("ffn-down (producer)", I)):
        ak = torch.randn(Mbig, K, device=dev).to(torch.bfloat16)
        w = (torch.randn(H, K, device=dev) * 0.03).to(torch.bfloat16)
        bias = torch.randn(H, device=dev) * 0.1
        g, bt = 1 + 0.1 * torch.randn(H, device=dev), 0.1 * torch.randn(H, device=dev)
        o1 = ops.gemm_ln_prod(ak, w, bias, x, st, g, bt, 1e-12, H)
        o2 = ops.gemm_ln_prod(ak[:Msmall].contiguous(), w, bias, x[:Msmall].contiguous(), st[:Msmall].contiguous(), g, bt, 1e-12, H)
        eq = [torch.equal(p[:Msmall], q) for p, q in zip(o1, o2)]
        d = (o1[0][:Msmall] - o2[0]).abs()
        bad_rows = (d.max(1).values > 0).nonzero().flatten()
        print("M %d vs %d  %-26s equal (f32, bf16, stats) %s  (differing rows: %d, first %s)" % (Mbig, Msmall, name, eq, bad_rows.numel(), bad_rows[:4].tolist()))
