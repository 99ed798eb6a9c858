R=$PWD
cd /tmp && export TMPDIR=/tmp
for w in 8 4; do
  rm -rf /tmp/kt$w; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt$w -- python $R/bench.py --steps 30 --warmup 10 --no-cpu --no-roofline --no-extra --no-sustained --tune 24=$w > /tmp/kt$w.json 2>/dev/null
  f=$(find /tmp/kt$w -name "*kernel_trace.csv" | head -1)
  echo "== waves $w: $(python -c "import json;d=json.load(open('/tmp/kt$w.json'));print(d['ms_per_step'])") ms/step under rocprof"
  python $R/tools/gap_analysis.py $f 800 | cut -c1-150
  s=$(find /tmp/kt$w -name "*kernel_stats.csv" | head -1)
  python - <<P
import csv
for r in csv.DictReader(open("$s")):
    n=r["Name"]
    if any(k in n for k in ("prod3","ffn_up","qkv3")): print("   %-40s calls %s avg %.2f us" % (n[:40], r["Calls"], float(r["AverageNs"])/1e3))
P
done
