python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/s3_gputest3.txt; cat gpurun_out/s3_gputest3.txt
for c in "--config cfg3 --agents 1" "--config cfg4 --agents 2" "--config cfg3" "--config cfg4"; do
for v in "" "W2C_S2WREG_FORM=0"; do
  echo "[$c $v]" >> gpurun_out/s3_ab7.txt
  env $v timeout 300 python bench.py $c --no-cpu-baseline --no-pmc --inflight 1 --steps 50 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])" >> gpurun_out/s3_ab7.txt 2>&1
done
done
cat gpurun_out/s3_ab7.txt
