#!/bin/bash
# usage: tools/r06/ab_inflight.sh "<env A>" "<env B>" [rounds]  -- interleaved bench.py runs with 3 forwards in flight
A="$1"; B="$2"; R="${3:-2}"
for r in $(seq 1 $R); do
  for v in "$A" "$B"; do
    env $v python bench.py --no-pmc --no-cpu-baseline --inflight 3 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
f=d.get('forwards_in_flight',{})
print('$v'.ljust(20), 'ms', d['ms_per_step'], 'inflight ms/forward', f.get('ms_per_forward'), f.get('value'), f.get('outputs_equal_headline'))"
  done
done
