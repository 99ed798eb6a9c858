#!/bin/bash
# Which limiter holds the shader clock down: amd-smi's throttle accumulators (PPT = package power tracking, thermal, PROCHOT) read before and after
# (a) 4000 steps of the fused encoder's bench step and (b) 3000 passes of the vendor chain (tools/yardstick.bin --chain); the share of the accumulation
# window spent in each violation is the accumulator's delta over ACCUMULATION_COUNTER's delta.   usage: tools/throttle_step_vs_chain.sh r04
T=$1; O=gpurun_out/${T}_throttle_step_vs_chain.txt
snap() { amd-smi metric -g 0 --throttle 2>/dev/null | grep -E "ACCUMULATION_COUNTER|PROCHOT_ACCUMULATED|PPT_ACCUMULATED|SOCKET_THERMAL_ACCUMULATED|VR_THERMAL_ACCUMULATED|HBM_THERMAL_ACCUMULATED" | tr -d ' ' | tr '\n' ' '; echo; }
: > $O
echo "idle      $(snap)" >> $O
python bench.py --steps 4000 --warmup 5 --no-cpu --no-extra --no-sustained --no-roofline > /tmp/ts_bench.json 2>/dev/null &
P=$!
sleep 9; echo "step@9s   $(snap)" >> $O
sleep 1.5; echo "step@10.5 $(snap)" >> $O
sleep 1.5; echo "step@12s  $(snap)" >> $O
wait $P
echo "after     $(snap)" >> $O
python3 -c "import json;d=json.loads(open('/tmp/ts_bench.json').read().strip().splitlines()[-1]);print('fused step:',d['ms_per_step'],'ms/step')" >> $O
sleep 3
echo "idle2     $(snap)" >> $O
tools/yardstick.bin --chain 3000 > /tmp/ts_chain.json 2>/dev/null &
P=$!
sleep 4; echo "chain@4s  $(snap)" >> $O
sleep 2; echo "chain@6s  $(snap)" >> $O
sleep 2; echo "chain@8s  $(snap)" >> $O
wait $P
echo "after2    $(snap)" >> $O
python3 -c "import json;d=json.load(open('/tmp/ts_chain.json'));print('vendor chain:',d['chain_ms_per_encoder_pass'],'ms per encoder pass')" >> $O
python3 - $O <<'PY' >> $O
import re, sys
rows = []
for line in open(sys.argv[1]):
    m = re.match(r"(\S+)\s+ACCUMULATION_COUNTER:(\d+) PROCHOT_ACCUMULATED:(\d+) PPT_ACCUMULATED:(\d+) SOCKET_THERMAL_ACCUMULATED:(\d+) VR_THERMAL_ACCUMULATED:(\d+) HBM_THERMAL_ACCUMULATED:(\d+)", line)
    if m:
        rows.append((m.group(1),) + tuple(int(x) for x in m.groups()[1:]))
print()
for a, b in zip(rows, rows[1:]):
    dc = b[1] - a[1]
    if dc > 0:
        print("%-10s -> %-10s window %9d ticks: PPT %.1f %%, socket thermal %.1f %%, VR thermal %.1f %%, HBM thermal %.1f %%, PROCHOT %.1f %%" %
              (a[0], b[0], dc, 100.0 * (b[3] - a[3]) / dc, 100.0 * (b[4] - a[4]) / dc, 100.0 * (b[5] - a[5]) / dc, 100.0 * (b[6] - a[6]) / dc, 100.0 * (b[2] - a[2]) / dc))
PY
cat $O
