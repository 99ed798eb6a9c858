"""Stand-alone A/B of the LayerNorm producers at the bench shape: row-major residual + slab epilogue (round 3 panel kernel) against the panel
residual + register-direct epilogue (round 5), both wave shapes.   python tools/rp_bench.py [--iters 200]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cpt_amd import ops, _lib as L  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--M", type=int, default=7680)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    M, H = a.M, 768
    x = torch.randn(M, H, device=dev) * 1.2 + 0.3
    hi, lo = ops.resid3_split(x)
    st = ops.row_stats_table(x)
    bias, g, bt = torch.randn(H, device=dev) * 0.1, 1 + torch.randn(H, device=dev) * 0.1, torch.randn(H, device=dev) * 0.1
    hp, lp = ops.panel_pack(hi), ops.panel_pack_bytes(lo)
    for K in (768, 3072):
        A = (torch.randn(M, K, device=dev)).to(torch.bfloat16)
        w = (torch.randn(H, K, device=dev) * 0.03).to(torch.bfloat16)
        apn = ops.panel_pack(A)
        for waves in (8, 4):
            res = {}
            for name, fn in (("slab", lambda: ops.gemm_ln_prod3_panel(apn, K, w, bias, hi, lo, st, g, bt, 1e-12, H, waves=waves)),
                             ("direct", lambda: ops.gemm_ln_prod3_rpanel(apn, K, w, bias, hp, lp, st, g, bt, 1e-12, H, waves=waves))):
                for _ in range(10):
                    fn()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.iters):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                res[name] = e0.elapsed_time(e1) / a.iters * 1e3
            print("K %4d waves %d  slab %.2f us  direct %.2f us  (includes ~3 torch allocs per call)" % (K, waves, res["slab"], res["direct"]), flush=True)
            # per-workgroup phase stamps (shader clock ticks; the kernels stamp start / after the prologue / after the K loop / end)
            import ctypes as C
            nwg = (M // 128) * (H // 192)
            for name, fn in (("slab", lambda: ops.gemm_ln_prod3_panel(apn, K, w, bias, hi, lo, st, g, bt, 1e-12, H, waves=waves)),
                             ("direct", lambda: ops.gemm_ln_prod3_rpanel(apn, K, w, bias, hp, lp, st, g, bt, 1e-12, H, waves=waves))):
                tr = torch.zeros(nwg * 8, dtype=torch.int64, device=dev)
                fn()
                L.lib().cpt_debug_gemm_trace(C.c_void_p(tr.data_ptr()))
                fn()
                torch.cuda.synchronize()
                L.lib().cpt_debug_gemm_trace(None)
                t = tr.view(nwg, 8).cpu()
                pro, kl, ep = ((t[:, 1] - t[:, 0]).float().mean().item(), (t[:, 2] - t[:, 1]).float().mean().item(), (t[:, 4] - t[:, 2]).float().mean().item())
                wall = ((t[:, 5] - t[:, 3]).float() * 0.01)
                print("        %-6s ticks per workgroup: prologue %.0f  K loop %.0f (%.0f per K-tile)  epilogue %.0f; workgroup wall %.2f us; first start -> last end %.2f us"
                      % (name, pro, kl, kl / (K // 64), ep, wall.mean().item(), (t[:, 5].max() - t[:, 3].min()).item() * 0.01), flush=True)


if __name__ == "__main__":
    main()
