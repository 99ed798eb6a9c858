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


if __name__ == "__main__":
    main()
