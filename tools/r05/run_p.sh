#!/bin/bash
out=gpurun_out/r05_p; mkdir -p $out
for q in 16 16 16 16 16 16 16 16 16 16 16 16 4 4 4 4 4 4; do
  GPU_MAX_HW_QUEUES=$q timeout 200 python bench.py --no-cpu-baseline --no-pmc --inflight 1 --steps 40 --warmup 10 --no-retry 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('Q=$q', d['ms_per_step'], d['config'].get('graph_audition_ms'))" | tee -a $out/summary.txt
done
