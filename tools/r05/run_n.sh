#!/bin/bash
out=gpurun_out/r05_n; mkdir -p $out
for q in 16 32 16 32; do for b in 3 2; do
  timeout 600 env GPU_MAX_HW_QUEUES=$q python tools/r05/hipgraph_oob_repro.py 12000 $b > $out/stress_q${q}_b$b.log 2>&1; echo "stress Q=$q branches=$b rc=$? $(tail -1 $out/stress_q${q}_b$b.log | cut -c1-80)" | tee -a $out/summary.txt
done; done
