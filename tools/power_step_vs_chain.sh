#!/bin/bash
# Package power and shader clock (rocm-smi every 0.2 s) under (a) the fused encoder's bench step and (b) the "vendor GEMM + row kernels" chain of
# tools/yardstick.bin --chain, each held for several seconds on the same box.   usage: tools/power_step_vs_chain.sh r04  -> gpurun_out/<tag>_power_step_vs_chain.txt
T=$1; R=$PWD; O=$R/gpurun_out/${T}_power_step_vs_chain.txt
sample() { while true; do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "sclk clock|Package Power" | tr '\n' ' '; echo; sleep 0.2; done; }
summ() { python3 - "$1" "$2" <<'PY'
import re, sys
pw, ck = [], []
for line in open(sys.argv[1]):
    m = re.search(r"sclk clock level: \S+ \((\d+)Mhz\).*Package Power \(W\): ([\d.]+)", line)
    if m and float(m.group(2)) > 600:          # samples while the load runs
        ck.append(int(m.group(1))); pw.append(float(m.group(2)))
if pw:
    print("%s: %d loaded samples, package power mean %.0f W (max %.0f), sclk mean %.0f MHz" % (sys.argv[2], len(pw), sum(pw) / len(pw), max(pw), sum(ck) / len(ck)))
else:
    print("%s: no loaded samples" % sys.argv[2])
PY
}
: > $O
sample > /tmp/ps_a.txt & S=$!
python bench.py --steps 4000 --warmup 5 --no-cpu --no-extra --no-sustained --no-roofline > /tmp/ps_bench.json 2>/dev/null
kill $S; wait $S 2>/dev/null
sleep 3
sample > /tmp/ps_b.txt & S=$!
tools/yardstick.bin --chain 3000 > /tmp/ps_chain.json 2>/dev/null
kill $S; wait $S 2>/dev/null
{ echo "## fused encoder step (bench.py --steps 4000): $(python3 -c "import json;d=json.loads(open('/tmp/ps_bench.json').read().strip().splitlines()[-1]);print(d['ms_per_step'],'ms/step,',d['value'],'pairs/s')")"
  summ /tmp/ps_a.txt "fused step"
  echo "## vendor chain (tools/yardstick.bin --chain 3000): $(python3 -c "import json;d=json.load(open('/tmp/ps_chain.json'));print(d['chain_ms_per_encoder_pass'],'ms per encoder pass')")"
  summ /tmp/ps_b.txt "vendor chain"
  echo; echo "## samples, fused step"; cat /tmp/ps_a.txt; echo; echo "## samples, vendor chain"; cat /tmp/ps_b.txt; } >> $O
head -5 $O
