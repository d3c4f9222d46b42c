#!/bin/bash
# Which clock / power does the chip hold while the forward loops?  Samples rocm-smi beside a long bench loop (no profiler attached).
# usage: tools/clock_watch.sh [env assignments ...]   (e.g. tools/clock_watch.sh W2C_REGH_WGS=256)
( env "$@" python bench.py --no-cpu-baseline --no-pmc --inflight 1 --steps 6000 --warmup 50 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', d['ms_per_step'], d['value'])" ) &
bp=$!
sleep 6
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|mclk|fclk" | tr -s ' ' | tr '\n' ';'
  echo
done
wait $bp
