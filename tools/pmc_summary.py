"""Per-kernel-family means of rocprofv3 --pmc counters collected over bench.py (tools/pmc_bench.sh).
python tools/pmc_summary.py <dir with the per-counter pass outputs> -> <dir>/summary.json"""
import collections
import csv
import glob
import json
import re
import sys


def family(name, prev):
    if "ffn_up_2pass_kernel" in name:
        return "gemm_ffn_up(+gelu)"                     # gemm_ffn.hip: the two-pass 384 x 256 FFN-up kernel
    if "qkv3_attn_kernel" in name:
        return "gemm_qkv_attn"                          # qkv_attn3.hip: QKV projection + attention, one workgroup per (sequence, three heads)
    if "prod3_panel_kernel" in name:                    # gemm_prod.hip: LayerNorm producers reading A from its panel copy
        m = re.search(r"prod3_panel_kernelILi\d+ELi\d+ELi(\d)E", name)      # round 4: the SITE template argument names the launch (0 attn-out, 1 FFN-down)
        if m:
            return "gemm_attn_out" if m.group(1) == "0" else "gemm_ffn_down"
        return "gemm_attn_out" if prev in ("attention", "gemm_qkv_attn") else "gemm_ffn_down"
    if "gemm_pipe_kernel" in name:
        # template arguments: <T, EPI, OT, ...>; EPI 0 none, 1 gelu, 3 resid, 6 / 11 LN producer, 7/8 LN consumer (+gelu), 9/10 fused QKV + attention
        if "Li9EDF16b" in name or "Li10EDF16b" in name:
            return "gemm_qkv_attn"                      # fused QKV projection + attention
        if "Li8EDF16b" in name or "Li1EDF16b" in name:
            return "gemm_ffn_up(+gelu)"
        if "Li7EDF16b" in name or "Li0EDF16b" in name:
            return "gemm_qkv"
        if "Li11Ef" in name or "Li6Ef" in name or "Li3Ef" in name:        # 11: LN producer with the 3-byte residual stream
            return "gemm_attn_out" if prev in ("attention", "gemm_qkv_attn") else "gemm_ffn_down"
        return "gemm_other"
    if "layernorm_rows" in name:
        return "layernorm"
    if "attention_kernel" in name:
        return "attention"
    return None


def main(d):
    res = {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        rows = list(csv.DictReader(open(f)))
        rows.sort(key=lambda r: int(r["Dispatch_Id"]))
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        prev, last_id, fam = None, None, None
        for r in rows:
            if r["Dispatch_Id"] != last_id:
                fam = family(r["Kernel_Name"], prev)
                if fam:
                    prev = fam
                last_id = r["Dispatch_Id"]
            if fam:
                acc[fam][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for fam, dd in acc.items():
            for c, v in dd.items():
                res.setdefault(fam, {})[c] = {"mean_per_launch": sum(v) / len(v), "launches": len(v)}
    out = json.dumps(res, indent=1, sort_keys=True)
    print(out)
    open(d + "/summary.json", "w").write(out)


if __name__ == "__main__":
    main(sys.argv[1])
