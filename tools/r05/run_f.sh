#!/bin/bash
# N full -m gpu suite runs, logs kept; stop early only on a crash (the log then holds the native backtrace)
tag=${1:-r05_f}; n=${2:-3}
out=gpurun_out/$tag; mkdir -p $out
for i in $(seq 1 $n); do
  timeout 1500 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider > $out/suite_$i.log 2>&1
  rc=$?
  echo "suite run $i rc=$rc : $(tail -1 $out/suite_$i.log | cut -c1-120)" | tee -a $out/summary.txt
done
