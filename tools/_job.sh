for i in 1 2; do
for v in "" "W2C_REGH_WGS=256" "W2C_REGH_FORM=7" "W2C_REGH_WGS=64"; do
  echo "[$v]" >> gpurun_out/s3_ab4.txt
  env $v timeout 300 python bench.py --no-cpu-baseline --no-pmc --inflight 1 --steps 50 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], [l['us_per_launch'] for l in d['roofline']['layers'] if l['cin']==64], d['config'].get('graph_audition_ms'))" >> gpurun_out/s3_ab4.txt 2>&1
done
done
for c in "--config cfg3 --agents 1" "--config cfg4 --agents 2"; do
for v in "" "W2C_REGH_WGS=256"; do
  echo "[$c $v]" >> gpurun_out/s3_ab4.txt
  env $v timeout 300 python bench.py $c --no-cpu-baseline --no-pmc --inflight 1 --steps 50 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])" >> gpurun_out/s3_ab4.txt 2>&1
done
done
cat gpurun_out/s3_ab4.txt
