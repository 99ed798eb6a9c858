"""Row-major LayerNorm producer (gemm.hip, A through the LDS ring) against the panel producer (gemm_prod.hip, A straight into
registers) at the bench shapes (GPU box only).  python tools/panel_bench.py [--batch 64] [--iters 50] [--rounds 3]"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpt_amd import _lib as L, ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--n", type=int, default=768, help="output columns (192: one column tile per row tile, 60 workgroups -- no sibling tiles share A rows)")
    ap.add_argument("--no-cold", action="store_true")
    ap.add_argument("--tune", default="", help="comma list of key=value for cpt_set_tuning (24=4: the 4-wave shape of the panel kernel)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    M, H, I = a.batch * 120, a.n, 3072
    torch.manual_seed(0)
    lib = L.lib()
    for kv in [t for t in a.tune.split(",") if t]:
        L.check(lib.cpt_set_tuning(int(kv.split("=")[0]), int(kv.split("=")[1])), "cpt_set_tuning")
    for name, K in (("attn_out", 768), ("ffn_down", I)):
        x = torch.randn(M, H, device=dev) * 1.2 + 0.3
        hi, lo = ops.resid3_split(x)
        st = ops.row_stats_table(x)
        act = torch.randn(M, K, device=dev).to(torch.bfloat16)
        w = (torch.randn(H, K, device=dev) * 0.03).to(torch.bfloat16)
        bias, g, bt = torch.randn(H, device=dev) * 0.1, 1 + torch.randn(H, device=dev) * 0.1, torch.randn(H, device=dev) * 0.1
        actp = ops.panel_pack(act)
        r = ops.gemm_ln_prod3(act, w, bias, hi, lo, st, g, bt, 1e-12, H)
        p = ops.gemm_ln_prod3_panel(actp, K, w, bias, hi, lo, st, g, bt, 1e-12, H)
        same = all(torch.equal(u, v) for u, v in zip(r, p))
        o_hi, o_lo, o_st = r
        s = L.stream_ptr()
        row = lambda: lib.cpt_gemm_ln_prod3(act.data_ptr(), K, w.data_ptr(), K, bias.data_ptr(), hi.data_ptr(), lo.data_ptr(), H, st.data_ptr(), g.data_ptr(),
                                            bt.data_ptr(), 1e-12, H, o_hi.data_ptr(), o_lo.data_ptr(), o_st.data_ptr(), H, M, H, K, s)
        pan = lambda: lib.cpt_gemm_ln_prod3_panel(actp.data_ptr(), w.data_ptr(), K, bias.data_ptr(), hi.data_ptr(), lo.data_ptr(), H, st.data_ptr(), g.data_ptr(),
                                                  bt.data_ptr(), 1e-12, H, o_hi.data_ptr(), o_lo.data_ptr(), o_st.data_ptr(), H, M, H, K, s)
        for rd in range(a.rounds):
            for tag, fn in (("row-major", row), ("panel", pan)):
                for _ in range(5):
                    fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.iters):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) / a.iters * 1e3
                print("%-9s M=%d N=%d K=%4d round %d %-9s %7.2f us  %7.1f TFLOP/s  bit-identical %s"
                      % (name, M, H, K, rd, tag, us, 2.0 * M * H * K / us / 1e6, same), flush=True)
        # per-workgroup phase stamps of the panel kernel (prologue / K loop / epilogue, shader clocks)
        nwg = (M // 128) * (H // 192)
        for abl in ((0, 2, 3) if a.tune else (0,)):       # timing experiments (results garbage unless 0): 1 half the A loads, 2 no A loads, 3 no W DMA
            L.check(lib.cpt_set_tuning(13, abl), "cpt_set_tuning")
            tr = torch.zeros(nwg * 8, dtype=torch.int64, device=dev)
            for _ in range(3):
                pan()
            lib.cpt_debug_gemm_trace(C.c_void_p(tr.data_ptr()))
            pan()
            torch.cuda.synchronize()
            lib.cpt_debug_gemm_trace(None)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                pan()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / a.iters * 1e3
            t = tr.view(nwg, 8).cpu()
            pro, kl, ep = (t[:, 1] - t[:, 0]).float().mean().item(), (t[:, 2] - t[:, 1]).float().mean().item(), (t[:, 4] - t[:, 2]).float().mean().item()
            dur = (t[:, 4] - t[:, 0]).float()
            span = (t[:, 5].max() - t[:, 3].min()).item() * 0.01          # us, from the 100 MHz counter
            late = (t[:, 3] - t[:, 3].min()).float() * 0.01
            wdur = (t[:, 5] - t[:, 3]).float() * 0.01
            ghz = (dur / wdur).mean().item() * 1e-3
            print("%-9s panel kernel abl %d: %.2f us; mean ticks per workgroup: prologue %.0f  K loop %.0f (%.0f per K-tile)  epilogue %.0f | whole: mean %.0f max %.0f "
                  "min %.0f; prologue = set-up %.0f + issue %.0f + wait; shader clock %.2f GHz; workgroup wall time mean %.2f max %.2f us; first start -> last end %.2f us; start spread mean %.2f max %.2f us"
                  % (name, abl, us, pro, kl, kl / (K // 64), ep, dur.mean().item(), dur.max().item(), dur.min().item(), t[:, 6].float().mean().item(), t[:, 7].float().mean().item(), ghz, wdur.mean().item(), wdur.max().item(),
                     span, late.mean().item(), late.max().item()), flush=True)
            if False:
                xcc = t[:, 6] & 15
                print("          per XCD (workgroups, mean duration, mean K loop): " + "  ".join(
                    "%d: %d %.0f %.0f" % (x, int((xcc == x).sum()), dur[xcc == x].mean().item(), (t[:, 2] - t[:, 1]).float()[xcc == x].mean().item()) for x in range(8)), flush=True)
        L.check(lib.cpt_set_tuning(13, 0), "cpt_set_tuning")
        if a.no_cold:
            continue
        # operand temperature, as inside the model: caches flushed by a 1 GiB fill, then chosen operands touched again
        big = torch.empty(1 << 28, dtype=torch.float32, device=dev)
        actp2 = actp.clone()
        for what in ("nothing warm", "A + residual warm, W from HBM", "W warm, A + residual cold", "all warm"):
            for rep in range(2):
                big.zero_()
                if "A + residual warm" in what or what == "all warm":
                    actp.copy_(actp2)
                    keep = (hi.clone(), lo.clone(), st.clone())
                if what.startswith("W warm") or what == "all warm":
                    keep2 = (w.clone(), bias.clone(), g.clone(), bt.clone())
                tr = torch.zeros(nwg * 8, dtype=torch.int64, device=dev)
                lib.cpt_debug_gemm_trace(C.c_void_p(tr.data_ptr()))
                pan()
                torch.cuda.synchronize()
                lib.cpt_debug_gemm_trace(None)
                t = tr.view(nwg, 8).cpu()
                dur = (t[:, 4] - t[:, 0]).float()
                wdur = (t[:, 5] - t[:, 3]).float() * 0.01
                pro, kl, ep = (t[:, 1] - t[:, 0]).float().mean().item(), (t[:, 2] - t[:, 1]).float().mean().item(), (t[:, 4] - t[:, 2]).float().mean().item()
                print("%-9s panel kernel after a cache flush, %-32s: prologue %5.0f  K loop %6.0f (%4.0f per K-tile)  epilogue %5.0f; shader clock %.2f GHz; "
                      "first start -> last end %.2f us" % (name, what, pro, kl, kl / (K // 64), ep, (dur / wdur).mean().item() * 1e-3,
                                                           (t[:, 5].max() - t[:, 3].min()).item() * 0.01), flush=True)
        del big


if __name__ == "__main__":
    main()
