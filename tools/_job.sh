for i in 1 2; do
for v in "" "W2C_VALUE_LDS_KB=82" "W2C_VALUE_LDS_KB=120" "W2C_POLICY_LDS_KB=82"; do
  echo "[$v]" >> gpurun_out/s3_ab10.txt
  env $v timeout 300 python bench.py --steps 100 --warmup 30 --no-cpu-baseline --no-pmc --inflight 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['config'].get('graph_audition_ms'))" >> gpurun_out/s3_ab10.txt 2>&1
done
done
for v in "" "W2C_VALUE_LDS_KB=82"; do echo "[$v]" >> gpurun_out/s3_ab10.txt; env $v timeout 200 python tools/chain_stamps.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/s3_ab10.txt; done
cat gpurun_out/s3_ab10.txt
