for i in 1 2; do
for v in "" "W2C_STEM_FORM=2" "W2C_STEM_WGS=128" "W2C_S2WREG_FORM=1" "W2C_WREG_MINCIN=128"; do
  echo "[$v]" >> gpurun_out/s3_ab5.txt
  env $v timeout 300 python bench.py --no-cpu-baseline --no-pmc --inflight 1 --steps 50 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['config'].get('graph_audition_ms'))" >> gpurun_out/s3_ab5.txt 2>&1
done
done
cat gpurun_out/s3_ab5.txt
