run() { python bench.py --no-cpu --no-io --no-live-pmc --no-extra --no-roofline --no-sustained "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
for b in 63 48 33; do
run --batch $b --tune 32=0
run --batch $b --tune 32=1
done
run --dtype bf16x3 --steps 10 --warmup 3 --tune 31=0
run --dtype bf16x3 --steps 10 --warmup 3 --tune 31=1
run --dtype fp32 --steps 10 --warmup 3 --tune 31=0
run --dtype fp32 --steps 10 --warmup 3 --tune 31=1
done
