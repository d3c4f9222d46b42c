#!/bin/bash
# usage: tools/r06/ab.sh "<env A>" "<env B>" [rounds] [extra bench args]   -- interleaved bench.py runs, prints ms_per_step (+ eager key)
A="$1"; B="$2"; R="${3:-2}"; shift 3
for r in $(seq 1 $R); do
  for v in "$A" "$B"; do
    env $v python bench.py --no-pmc --no-cpu-baseline --inflight 1 "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v'.ljust(28), 'ms', d['ms_per_step'], 'eager', d.get('eager',{}).get('ms_per_step'), 'evalpath', d.get('evaluator_path',{}).get('ms_per_step'), 'first', d.get('first_forward_ms'), 'host', d['host_enqueue']['median_ms'], 'frac', d['roofline']['frac'], d['config']['launch'][:40], 'retry' if 'retry' in d else '')"
  done
done
