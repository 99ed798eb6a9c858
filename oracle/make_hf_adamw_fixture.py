"""TEST INFRASTRUCTURE (oracle side): golden vectors for the pytorch_transformers.AdamW arithmetic (the optimizer of the GQA / VCR few-shot drivers,
/root/reference/Oscar/oscar/fewshot/gqa_cpt.py:342-348, vcr_nsp_cpt.py:385-386).

Its source (huggingface/transformers @ 067923d, optimization.py) is NOT under /root/reference, and the installed transformers (5.15) no longer ships an
AdamW, so there is nothing to execute.  VERDICT r5 item 9 asks for the next best pin: an INDEPENDENT second restatement.  This script is one -- NumPy
float64, written from the papers' form of the algorithm (Kingma & Ba's bias-corrected moments m^ = m / (1 - b1^t), v^ = v / (1 - b2^t); Loshchilov &
Hutter's decoupled decay) with that implementation's two placements:
  * eps is added to sqrt(v), NOT to sqrt(v^):  p -= lr * m^ / (sqrt(v^) + eps / sqrt(1 - b2^t))      (<=> step_size = lr sqrt(1 - b2^t) / (1 - b1^t), denom = sqrt(v) + eps)
  * the decay acts on the UPDATED parameter:   p -= lr * wd * p
  * correct_bias = False: m^ = m, v^ = v, eps as is.
It shares no code with oracle/cpt_oracle.py:adamw_step_hf (float32 torch, the in-place call sequence of the implementation);
tests/test_host_cpu.py::test_hf_adamw_restatements_agree holds the two against each other on the vectors written here
(tests/golden/hf_adamw.npz).  Run in the build container:  python oracle/make_hf_adamw_fixture.py"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "hf_adamw.npz")


def hf_adamw_f64(p, g, m, v, t, lr, b1, b2, eps, wd, correct_bias=True):
    m = b1 * m + (1.0 - b1) * g
    v = b2 * v + (1.0 - b2) * np.square(g)
    if correct_bias:
        c1, c2 = 1.0 - b1 ** t, 1.0 - b2 ** t
        m_hat, v_hat, eps_hat = m / c1, v / c2, eps / np.sqrt(c2)
    else:
        m_hat, v_hat, eps_hat = m, v, eps
    p = p - lr * m_hat / (np.sqrt(v_hat) + eps_hat)
    p = p - lr * wd * p
    return p, m, v


def main():
    rng = np.random.default_rng(20240607)
    n, steps = 512, 6
    out = {}
    for tag, (b1, b2, eps, wd, cb) in {"gqa": (0.9, 0.999, 1e-8, 0.05, True),        # gqa_cpt.py:342-348 (adam_epsilon 1e-8, weight_decay 0.05)
                                       "vcr": (0.9, 0.999, 1e-6, 0.01, True),
                                       "nobias": (0.9, 0.98, 1e-6, 0.01, False)}.items():
        p = rng.standard_normal(n) * 0.02
        p[: n // 8] *= 50.0                     # a few large weights (decay term visible)
        m = np.zeros(n)
        v = np.zeros(n)
        g_all = rng.standard_normal((steps, n)) * np.exp(rng.uniform(-9, 1, size=(1, n)))      # gradients over ten orders of magnitude
        g_all[:, n // 2: n // 2 + 64] = 0.0     # parameters with a zero gradient (update = -lr wd p only)
        lrs = 5e-5 * np.array([0.2, 0.6, 1.0, 0.8, 0.6, 0.4])
        out[tag + "_p0"] = p.copy()
        out[tag + "_g"] = g_all
        out[tag + "_lr"] = lrs
        out[tag + "_hyper"] = np.array([b1, b2, eps, wd, 1.0 if cb else 0.0])
        ps, ms, vs = [], [], []
        for t in range(1, steps + 1):
            p, m, v = hf_adamw_f64(p, g_all[t - 1], m, v, t, lrs[t - 1], b1, b2, eps, wd, cb)
            ps.append(p.copy()); ms.append(m.copy()); vs.append(v.copy())
        out[tag + "_p"] = np.stack(ps)
        out[tag + "_m"] = np.stack(ms)
        out[tag + "_v"] = np.stack(vs)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
