"""Render tools/kloop_diag.sh's passes as the markdown table VERDICT r3 item 1 asks for (profiles/r04_kloop_vs_hipblaslt.md).
    python tools/kloop_diag_md.py <dir> > out.md"""
import collections
import csv
import glob
import json
import os
import re
import sys


def main_kernel_rows(d, hints):
    """counter rows of the loop's dominant kernel in one pass directory: {counter: [values]}, kernel name, durations"""
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    best = None
    for k, dd in acc.items():
        if not any(h in k for h in hints):
            continue
        n = max(len(v) for v in dd.values())
        if best is None or n > best[1]:
            best = (k, n)
    if best is None:
        return None, {}
    return best[0], {c: sum(v) / len(v) for c, v in acc[best[0]].items()}


def kt_mean(d, name_hint):
    for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if name_hint and name_hint[:60] in r["Name"]:
                return float(r["AverageNs"]) / 1e3, int(r["Calls"])
    return None, 0


def power(fn):
    w, clk = [], []
    try:
        for line in open(fn):
            m = re.search(r"Package Power \(W\): ([0-9.]+)", line)
            c = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", line)
            if m and float(m.group(1)) > 600:
                w.append(float(m.group(1)))
                if c:
                    clk.append(float(c.group(1)))
    except Exception:
        pass
    return (sum(w) / len(w) if w else None, max(w) if w else None, sum(clk) / len(clk) if clk else None, len(w))


def main(O):
    out = {}
    for sn, K in (("ffn_down", 3072), ("attn_out", 768)):
        for who in ("hbl", "ours"):
            ctr, name = {}, None
            for d in sorted(glob.glob("%s/%s_%s_p*/" % (O, who, sn))):
                k, c = main_kernel_rows(d, ("Cijk_",) if who == "hbl" else ("prod3_panel", "gemm_pipe_kernel"))
                if k:
                    name = name or k
                    ctr.update(c)
            us, calls = kt_mean("%s/%s_%s_kt" % (O, who, sn), name)
            pw = power("%s/power_%s_%s.txt" % (O, who, sn))
            out["%s/%s" % (sn, who)] = {"kernel": name, "us_kernel_trace": us, "calls": calls, "counters": ctr,
                                       "power_mean_W": pw[0], "power_max_W": pw[1], "sclk_mean_MHz": pw[2], "power_samples": pw[3]}
    json.dump(out, open(O + "/summary.json", "w"), indent=1, sort_keys=True)
    print("# K loop: hipBLASLt's winner against the fused LayerNorm producer (rocprofv3 --pmc, separate passes; tools/kloop_diag.sh)\n")
    print("M = 7680, N = 768; bf16 operands uniform in [-1, 1); counters are means per launch over the loop's launches; `per MFMA` divides by")
    print("SQ_INSTS_MFMA when the counter exists, else by the MFMA count the shape implies.\n")
    for sn, K in (("ffn_down", 3072), ("attn_out", 768)):
        a, b = out["%s/hbl" % sn], out["%s/ours" % sn]
        print("## %s (K = %d)\n" % (sn, K))
        print("| | hipBLASLt winner | ours (`prod3_panel_kernel`, fused epilogue) |\n|---|---|---|")
        print("| kernel | `%s` | `%s` |" % ((a["kernel"] or "?")[:110], (b["kernel"] or "?")[:80]))
        print("| us per launch (kernel trace) | %s | %s |" % (a["us_kernel_trace"], b["us_kernel_trace"]))
        print("| package power mean / max (W), sclk mean (MHz) | %s / %s, %s | %s / %s, %s |" % (a["power_mean_W"], a["power_max_W"], a["sclk_mean_MHz"], b["power_mean_W"], b["power_max_W"], b["sclk_mean_MHz"]))
        keys = sorted(set(a["counters"]) | set(b["counters"]))
        for k in keys:
            va, vb = a["counters"].get(k), b["counters"].get(k)
            print("| %s | %s | %s |" % (k, "%.4g" % va if va is not None else "-", "%.4g" % vb if vb is not None else "-"))
        # derived
        flops = 2.0 * 7680 * 768 * K

        def per_mfma(c, who):
            v = who["counters"].get(c)
            n = who["counters"].get("SQ_INSTS_MFMA") or who["counters"].get("SQ_INSTS_VALU_MFMA_MOPS_BF16")
            return (v / n) if (v is not None and n) else None
        for c in ("SQ_INSTS_LDS", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT"):
            pa, pb = per_mfma(c, a), per_mfma(c, b)
            print("| %s per SQ_INSTS_MFMA | %s | %s |" % (c, "%.3f" % pa if pa is not None else "-", "%.3f" % pb if pb is not None else "-"))
        for who, tag in ((a, "hipBLASLt"), (b, "ours")):
            m, bz = who["counters"].get("SQ_VALU_MFMA_BUSY_CYCLES"), who["counters"].get("SQ_BUSY_CYCLES")
            if m and bz:
                print("| MFMA busy / SQ busy (%s) | %.3f | |" % (tag, m / bz) if who is a else "| MFMA busy / SQ busy (%s) | | %.3f |" % (tag, m / bz))
        print()


if __name__ == "__main__":
    main(sys.argv[1])
