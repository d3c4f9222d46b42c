#!/bin/bash
# round 5: hunt the hipGraphLaunch SIGSEGV with the native backtrace handler; packed-f32 op_sel probe; one-graph sharded runs
out=gpurun_out/r05_e; mkdir -p $out
timeout 300 tools/ubench/bin/pkfma > $out/pkfma_probe.txt 2>&1
ok=0
for i in 1 2 3 4 5; do
  timeout 300 python bench.py --gpus 1 --force-sharded --batch 2 --size 128 --steps 5 --warmup 2 --no-pmc > $out/fs_$i.json 2> $out/fs_$i.err
  if [ $? -eq 0 ] && grep -q "one hip-graph" $out/fs_$i.json; then ok=$((ok+1)); rm -f $out/fs_$i.err; fi
done
echo "force-sharded one-graph runs ok: $ok / 5" | tee -a $out/summary.txt
for i in $(seq 1 ${1:-10}); do
  timeout 600 python -m pytest tests/test_forward_gpu.py -x -q -m gpu -p no:cacheprovider -k "several_engines" > $out/eng_$i.log 2>&1
  rc=$?; echo "engines-in-flight $i rc=$rc $(tail -1 $out/eng_$i.log)" | tee -a $out/summary.txt
  [ $rc -eq 0 ] && rm -f $out/eng_$i.log
done
for i in $(seq 1 ${2:-3}); do
  timeout 1200 python -m pytest tests/test_forward_gpu.py tests/test_srms.py -x -q -m gpu -p no:cacheprovider > $out/pair_$i.log 2>&1
  rc=$?; echo "forward+srms $i rc=$rc $(tail -1 $out/pair_$i.log)" | tee -a $out/summary.txt
done
