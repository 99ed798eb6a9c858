"""Launch one GEMM shape a few times (for rocprofv3 --pmc runs).  python tools/gemm_prof.py qkv 3"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cpt_amd import _lib as L, ops  # noqa: E402

name, variant = sys.argv[1], int(sys.argv[2])
dt = torch.bfloat16
M, H, I = 64 * 120, 768, 3072
shapes = {"qkv": (M, 3 * H, H, L.EPI_NONE, dt), "attn_out": (M, H, H, L.EPI_RESID, torch.float32),
          "ffn_up": (M, I, H, L.EPI_GELU, dt), "ffn_down": (M, H, I, L.EPI_RESID, torch.float32)}
m, n, k, epi, odt = shapes[name]
dev = torch.device("cuda:0")
torch.manual_seed(0)
x = torch.randn(m, k, device=dev).to(dt)
w = (torch.randn(n, k, device=dev) * 0.05).to(dt)
b = torch.randn(n, device=dev)
r = torch.randn(m, n, device=dev) if epi == L.EPI_RESID else None
L.check(L.lib().cpt_set_tuning(0, variant))
for _ in range(5):
    ops.gemm(x, w, b, epi=epi, resid=r, out_dtype=odt)
torch.cuda.synchronize()
