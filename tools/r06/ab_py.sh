#!/bin/bash
# A/B of two versions of the package's Python files: tools/r06/ab_py.sh <basedir> [rounds] [extra bench args] -- <basedir>/*.py ("base") swapped
# over multiagentperception_amd/*.py against the tree's ("new"), interleaved bench.py runs
BASE="$1"; R="${2:-3}"; shift 2
PK=multiagentperception_amd
mkdir -p /tmp/w2c_new_py; for f in $BASE/*.py; do cp $PK/$(basename $f) /tmp/w2c_new_py/; done
for r in $(seq 1 $R); do
  for v in base new; do
    if [ $v = base ]; then cp $BASE/*.py $PK/; else cp /tmp/w2c_new_py/*.py $PK/; fi
    python bench.py --no-pmc --no-cpu-baseline --inflight 1 "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v'.ljust(6), 'ms', d['ms_per_step'], 'eager', d.get('eager',{}).get('ms_per_step'), 'evalpath', d.get('evaluator_path',{}).get('ms_per_step'), 'host', d['host_enqueue']['median_ms'], 'frac', d['roofline']['frac'])"
  done
done
cp /tmp/w2c_new_py/*.py $PK/
