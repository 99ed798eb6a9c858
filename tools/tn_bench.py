"""Timing of the TN (weight-gradient) GEMM on the training step's shapes (GPU box).  python tools/tn_bench.py [rows]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpt_amd import ops  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 3840
dev = torch.device("cuda:0")
torch.manual_seed(0)
for name, M, N in (("qkv", 2304, 768), ("attn_out", 768, 768), ("ffn_up", 3072, 768), ("ffn_down", 768, 3072), ("img", 768, 2112)):
    K = rows if name != "img" else rows // 120 * 50 // 64 * 64
    a = torch.randn(K, M, device=dev).to(torch.bfloat16)
    w = torch.randn(K, N, device=dev).to(torch.bfloat16)
    for scratch in (True, False):
        for _ in range(3):
            ops.gemm_tn(a, w, scratch)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.gemm_tn(a, w, scratch)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print("%-9s M=%4d N=%4d K=%5d split-K %-3s %7.1f us  %6.1f TFLOP/s" % (name, M, N, K, "yes" if scratch else "no", us, 2.0 * M * N * K / us / 1e6))
