cd $GRAFT_REPO_ROOT
run() { python bench.py "$@" --steps 30 --no-pmc --no-cpu-baseline --inflight 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', '|', d['ms_per_step'], d['value'], d['roofline']['frac'], d['config']['launch'])"; }
run --config cfg3 --agents 1 --force-sharded
W2C_SHARD_ONE_GRAPH=0 run --config cfg3 --agents 1 --force-sharded
run --config cfg3 --agents 1
run --config cfg4 --agents 2 --force-sharded
run --config cfg4 --agents 2
run --config cfg2 --force-sharded
W2C_SHARD_ONE_GRAPH=0 run --config cfg2 --force-sharded
run --config cfg2
