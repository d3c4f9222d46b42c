cd $GRAFT_REPO_ROOT
B="python bench.py --steps 50 --warmup 5 --no-pmc --no-cpu-baseline --inflight 1"
for i in 1 2 3; do
  for v in 0 1 2; do
    env W2C_VALUE_LAG=$v $B 2>/dev/null | python -c "import sys,json;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('W2C_VALUE_LAG=$v',d['ms_per_step'])"
  done
done
