for i in 1 2; do
for v in "--steps 20 --warmup 3" "--steps 100 --warmup 20" "--steps 200 --warmup 50" "--steps 1000 --warmup 200"; do
  echo "[$v]" >> gpurun_out/s3_ab8.txt
  timeout 300 python bench.py $v --no-cpu-baseline --no-pmc --inflight 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['config'].get('graph_audition_ms'))" >> gpurun_out/s3_ab8.txt 2>&1
done
done
for i in 1 2; do
for v in "" "W2C_UPS_LDS_KB=40" "W2C_UPS_LDS_KB=64"; do
  echo "[$v]" >> gpurun_out/s3_ab8.txt
  env $v timeout 300 python bench.py --steps 200 --warmup 50 --no-cpu-baseline --no-pmc --inflight 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['config'].get('graph_audition_ms'))" >> gpurun_out/s3_ab8.txt 2>&1
done
done
cat gpurun_out/s3_ab8.txt
