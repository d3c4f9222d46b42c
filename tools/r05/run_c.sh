#!/bin/bash
# full GPU suite N times (logs kept), then a bench line
tag=${1:-r05_c}; n=${2:-1}
out=gpurun_out/$tag; mkdir -p $out
for i in $(seq 1 $n); do
  timeout 1500 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider --durations=12 > $out/suite_$i.log 2>&1
  echo "suite run $i rc=$? : $(tail -1 $out/suite_$i.log)" | tee -a $out/summary.txt
done
timeout 600 python bench.py > $out/bench.json 2> $out/bench.err
echo "bench rc=$?" | tee -a $out/summary.txt
cat $out/bench.json | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    print({k: d[k] for k in ('value', 'ms_per_step') if k in d}, d.get('roofline', {}).get('frac'))
" | tee -a $out/summary.txt
