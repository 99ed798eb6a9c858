"""One stand-alone producer launch of the fused encoder in a loop (GPU box only), the counterpart of `tools/yardstick.bin --loop`
for rocprofv3 --pmc / --kernel-trace passes and power sampling (tools/kloop_diag.sh).
    python tools/kloop_ours.py --shape ffn_down|attn_out [--iters 30] [--seconds 0] [--kernel panel|rowmajor]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpt_amd import _lib as L, ops  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="ffn_down", choices=["ffn_down", "attn_out"])
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--seconds", type=float, default=0.0)
    ap.add_argument("--kernel", default="panel", choices=["panel", "rowmajor"])
    ap.add_argument("--tune", default="")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    M, H = 64 * 120, 768
    K = 3072 if a.shape == "ffn_down" else 768
    torch.manual_seed(0)
    lib = L.lib()
    for kv in [t for t in a.tune.split(",") if t]:
        k, v = kv.split("=")
        L.check(lib.cpt_set_tuning(int(k), int(v)))
    x = torch.randn(M, H, device=dev) * 1.2 + 0.3
    hi, lo = ops.resid3_split(x)
    st = ops.row_stats_table(x)
    act = (torch.rand(M, K, device=dev) * 2 - 1).to(torch.bfloat16)           # uniform [-1, 1) like tools/yardstick.hip
    w = (torch.rand(H, K, device=dev) * 2 - 1).to(torch.bfloat16)
    bias, g, bt = torch.randn(H, device=dev) * 0.1, 1 + torch.randn(H, device=dev) * 0.1, torch.randn(H, device=dev) * 0.1
    actp = ops.panel_pack(act)
    o_hi, o_lo, o_st = ops.gemm_ln_prod3(act, w, bias, hi, lo, st, g, bt, 1e-12, H)
    s = L.stream_ptr()
    if a.kernel == "panel":
        fn = lambda: lib.cpt_gemm_ln_prod3_panel(actp.data_ptr(), w.data_ptr(), K, bias.data_ptr(), hi.data_ptr(), lo.data_ptr(), H, st.data_ptr(), g.data_ptr(),
                                                 bt.data_ptr(), 1e-12, H, o_hi.data_ptr(), o_lo.data_ptr(), o_st.data_ptr(), H, M, H, K, s)
    else:
        fn = lambda: lib.cpt_gemm_ln_prod3(act.data_ptr(), K, w.data_ptr(), K, bias.data_ptr(), hi.data_ptr(), lo.data_ptr(), H, st.data_ptr(), g.data_ptr(),
                                           bt.data_ptr(), 1e-12, H, o_hi.data_ptr(), o_lo.data_ptr(), o_st.data_ptr(), H, M, H, K, s)
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / a.iters * 1e3
    spins = 0
    if a.seconds > 0:
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < a.seconds:
            for _ in range(200):
                fn()
            torch.cuda.synchronize()
            spins += 200
    print('{"shape": "%s", "kernel": "%s", "M": %d, "N": %d, "K": %d, "loop_us": %.2f, "TFLOPs": %.1f, "sustained_launches": %d}'
          % (a.shape, a.kernel, M, H, K, us, 2.0 * M * H * K / us / 1e6, spins), flush=True)


if __name__ == "__main__":
    main()
