"""A/B timing of the GEMM variants on the hot path's shapes (GPU box only).
python tools/gemm_bench.py [--batch 64] [--dtype bf16]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpt_amd import _lib as L, ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--variants", default="0,1,2")
    ap.add_argument("--iters", type=int, default=30)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    M, H, I = a.batch * 120, 768, 3072
    shapes = [("qkv", M, 3 * H, H, L.EPI_NONE, dt), ("attn_out", M, H, H, L.EPI_RESID, torch.float32),
              ("ffn_up", M, I, H, L.EPI_GELU, dt), ("ffn_down", M, H, I, L.EPI_RESID, torch.float32),
              ("decoder", a.batch, 30522, H, L.EPI_NONE, torch.float32)]
    torch.manual_seed(0)
    for name, m, n, k, epi, odt in shapes:
        x = torch.randn(m, k, device=dev).to(dt)
        w = (torch.randn(n, k, device=dev) * 0.05).to(dt)
        b = torch.randn(n, device=dev)
        r = torch.randn(m, n, device=dev) if epi == L.EPI_RESID else None
        ref = None
        for v in [int(t) for t in a.variants.split(",")]:
            L.check(L.lib().cpt_set_tuning(0, v))
            out = ops.gemm(x, w, b, epi=epi, resid=r, out_dtype=odt)
            torch.cuda.synchronize()
            if ref is None:
                ref = out.float()
            err = (out.float() - ref).abs().max().item()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(5):
                ops.gemm(x, w, b, epi=epi, resid=r, out_dtype=odt)
            e0.record()
            for _ in range(a.iters):
                ops.gemm(x, w, b, epi=epi, resid=r, out_dtype=odt)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / a.iters
            print("%-9s %5dx%5dx%4d variant %d: %8.2f us  %7.1f TFLOP/s  (max diff vs first variant %.2e)"
                  % (name, m, n, k, v, ms * 1e3, 2.0 * m * n * k / ms / 1e9, err), flush=True)
    L.check(L.lib().cpt_set_tuning(0, 1))


if __name__ == "__main__":
    main()
