"""Per-workgroup phase timeline of the pipelined GEMM (shader-clock stamps).  GPU box only."""
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
abls = [int(v) for v in (sys.argv[3].split(",") if len(sys.argv) > 3 else ["0"])]
for variant, abl in [(int(v), a) for v in sys.argv[1].split(",") for a in abls]:
    L.check(L.lib().cpt_set_tuning(1, abl))
    for name in sys.argv[2].split(","):
        m, n, k, epi, odt = shapes[name]
        torch.manual_seed(0)
        x = torch.randn(m, k, device=dev).to(dt)
        w = (torch.randn(n, k, device=dev) * 0.05).to(dt)
        b = torch.randn(n, device=dev)
        r = torch.randn(m, n, device=dev) if epi == L.EPI_RESID else None
        L.check(L.lib().cpt_set_tuning(0, variant))
        for _ in range(3):
            ops.gemm(x, w, b, epi=epi, resid=r, out_dtype=odt)
        buf = torch.zeros(4096 * 8, device=dev, dtype=torch.int64)
        L.lib().cpt_debug_gemm_trace(buf.data_ptr())
        ops.gemm(x, w, b, epi=epi, resid=r, out_dtype=odt)
        torch.cuda.synchronize()
        L.lib().cpt_debug_gemm_trace(None)
        t = buf.cpu().numpy().reshape(-1, 8)
        t = t[t[:, 0] != 0]
        t0 = t[:, 0].min()
        pro, loop, stg, epi_t = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 4] - t[:, 3]
        tot = t[:, 4] - t[:, 0]
        print("abl %d variant %d %-8s wgs=%4d  span=%7d ticks | per-WG mean: prologue %6.0f  k-loop %6.0f  stage %6.0f  epilogue %6.0f  total %6.0f"
              % (abl, variant, name, len(t), t[:, 4].max() - t0, pro.mean(), loop.mean(), stg.mean(), epi_t.mean(), tot.mean()))
        starts = np.sort(t[:, 0] - t0)
        print("   start times pct [0,25,50,75,100]:", np.percentile(starts, [0, 25, 50, 75, 100]).astype(int),
              " k-loop p10/p90: %d/%d" % (np.percentile(loop, 10), np.percentile(loop, 90)))
        print("   inside k-loop (wave 0, sums over tiles): vmcnt-wait %6.0f  barrier %6.0f  glds-issue %6.0f  rest(ds_read+mfma) %6.0f"
              % (t[:, 5].mean(), t[:, 6].mean(), t[:, 7].mean(), (loop - t[:, 5] - t[:, 6] - t[:, 7]).mean()))
L.check(L.lib().cpt_set_tuning(0, 3))
L.check(L.lib().cpt_set_tuning(1, 0))
