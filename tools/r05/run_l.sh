#!/bin/bash
out=gpurun_out/r05_l; mkdir -p $out
run() { name=$1; shift; echo "=== $name" | tee -a $out/summary.txt; ( timeout 300 env "$@" python tools/r05/hipgraph_oob_repro.py 600 2 > $out/$name.log 2>&1; echo "rc=$?" >> $out/$name.log ); tail -2 $out/$name.log | tee -a $out/summary.txt; grep -m1 "libamdhip64" $out/$name.log | tee -a $out/summary.txt; }
run default_a X=1
run default_b X=1
run default_c X=1
run hwq8 GPU_MAX_HW_QUEUES=8
run hwq16 GPU_MAX_HW_QUEUES=16
run hwq16_b GPU_MAX_HW_QUEUES=16
