"""LayerNorm-consumer GEMM (FFN-up / stand-alone QKV projection) stand-alone: the two-pass kernel (gemm_ffn.hip, variant 20) against the
4-wave kernel (gemm_ffn4.hip, variant 21), with the 4-wave kernel's per-workgroup phase stamps (GPU box only).
    python tools/cons_bench.py [--m 7680] [--iters 100]"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpt_amd import _lib as L, ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=7680)
    ap.add_argument("--iters", type=int, default=100)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = L.lib()
    torch.manual_seed(0)
    M, K = a.m, 768
    for N, gelu in ((3072, True), (2304, False)):
        x = torch.randn(M, K, device=dev) * 1.3 + 0.4
        act = x.to(torch.bfloat16)
        st = ops.row_stats_table(x)
        wf = (torch.randn(N, K, device=dev) * 0.04).to(torch.bfloat16)
        colc = wf.float().sum(1).contiguous()
        cold = torch.randn(N, device=dev) * 0.1
        outs = {}
        for v in (20, 21, 20, 21):
            L.check(lib.cpt_set_tuning(0, v))
            fn = lambda: ops.gemm_ln_cons(act, wf, st, colc, cold, 1e-12, K, gelu)
            outs[v] = fn()
            for _ in range(5):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / a.iters * 1e3
            msg = "N=%d gelu=%d variant %d: %.2f us  %.1f TFLOP/s" % (N, gelu, v, us, 2.0 * M * N * K / us / 1e6)
            if v == 21:
                nwg = ((M + 191) // 192) * (N // 256)
                tr = torch.zeros(nwg * 8, dtype=torch.int64, device=dev)
                lib.cpt_debug_gemm_trace(C.c_void_p(tr.data_ptr()))
                fn()
                torch.cuda.synchronize()
                lib.cpt_debug_gemm_trace(None)
                t = tr.view(nwg, 8).cpu()
                pro, kl, ep = (t[:, 1] - t[:, 0]).float().mean().item(), (t[:, 2] - t[:, 1]).float().mean().item(), (t[:, 4] - t[:, 2]).float().mean().item()
                msg += " | ticks per workgroup: prologue %.0f  K loop %.0f (%.0f per K-tile)  epilogue %.0f" % (pro, kl, kl / (K // 64), ep)
            print(msg, flush=True)
        L.check(lib.cpt_set_tuning(0, 3))
        print("   bit-identical:", torch.equal(outs[20], outs[21]))


if __name__ == "__main__":
    main()
