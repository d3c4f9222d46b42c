#!/bin/bash
# round 5: suite stability (N full -m gpu runs) + the sharded one-graph step K times in fresh processes (watchdog observation)
tag=${1:-r05_d}; n=${2:-2}; k=${3:-25}
out=gpurun_out/$tag; mkdir -p $out
for i in $(seq 1 $n); do
  timeout 1500 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider > $out/suite_$i.log 2>&1
  echo "suite run $i rc=$? : $(tail -1 $out/suite_$i.log)" | tee -a $out/summary.txt
done
ok=0
for i in $(seq 1 $k); do
  timeout 300 python bench.py --gpus 1 --force-sharded --batch 2 --size 128 --steps 5 --warmup 2 --no-pmc > $out/fs_$i.json 2> $out/fs_$i.err
  rc=$?
  if [ $rc -eq 0 ] && grep -q "one hip-graph" $out/fs_$i.json; then ok=$((ok+1)); rm -f $out/fs_$i.err; else echo "force-sharded run $i rc=$rc" | tee -a $out/summary.txt; fi
done
echo "force-sharded one-graph runs ok: $ok / $k" | tee -a $out/summary.txt
