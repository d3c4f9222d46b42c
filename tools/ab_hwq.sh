# GPU_MAX_HW_QUEUES (ROCm's cap on hardware queues per process, default 4) against the headline, the three-forwards-in-flight side figure
# and the evaluator path; interleaved, fresh process each
cd $GRAFT_REPO_ROOT
for i in 1 2; do
  for q in ${QS:-1 2 3 4 5 6}; do
    GPU_MAX_HW_QUEUES=$q python bench.py --steps 40 --warmup 5 --no-pmc --no-cpu-baseline 2>/dev/null | python -c "import sys,json;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('GPU_MAX_HW_QUEUES=$q',d['ms_per_step'], d['forwards_in_flight']['ms_per_forward'], d['evaluator_path']['ms_per_step'])"
  done
done
