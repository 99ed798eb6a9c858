for w in 8 4; do echo "== waves $w"; python tools/panel_bench.py --rounds 1 --iters 100 --no-cold --tune 24=$w 2>&1 | grep -E "panel kernel abl|panel " ; done
