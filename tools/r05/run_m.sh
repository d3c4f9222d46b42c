#!/bin/bash
out=gpurun_out/r05_m; mkdir -p $out
for q in 8 16; do for b in 2 3 4; do
  timeout 300 env GPU_MAX_HW_QUEUES=$q python tools/r05/hipgraph_oob_repro.py 2000 $b > $out/stress_q${q}_b$b.log 2>&1; echo "stress Q=$q branches=$b rc=$? $(tail -1 $out/stress_q${q}_b$b.log | cut -c1-80)" | tee -a $out/summary.txt
done; done
for q in 4 8 16 4 8 16; do
  GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --no-cpu-baseline --no-pmc --inflight 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('GPU_MAX_HW_QUEUES=$q', d['ms_per_step'], d['value'], d.get('graph_audition_ms'), d.get('retry'))" | tee -a $out/summary.txt
done
