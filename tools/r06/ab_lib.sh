#!/bin/bash
# A/B of two builds of the library: tools/r06/ab_lib.sh <base.so> [rounds]  -- the tree's lib/libw2c_hip.so ("new") against <base.so>, interleaved
BASE="$1"; R="${2:-3}"
NEW=multiagentperception_amd/lib/libw2c_hip.so
cp $NEW /tmp/w2c_new.so
for r in $(seq 1 $R); do
  for v in base new; do
    if [ $v = base ]; then cp "$BASE" $NEW; else cp /tmp/w2c_new.so $NEW; fi
    python bench.py --no-pmc --no-cpu-baseline --inflight 1 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v'.ljust(6), 'ms', d['ms_per_step'], 'eager', d.get('eager',{}).get('ms_per_step'), 'evalpath', d.get('evaluator_path',{}).get('ms_per_step'), 'frac', d['roofline']['frac'])"
  done
done
cp /tmp/w2c_new.so $NEW
