# single-rank proxies of cfg 3 / cfg 4 (profiles/r03_rank_shapes.txt): one rank's share through the sharded path and as a plain forward
cd $GRAFT_REPO_ROOT
run() { python bench.py "$@" --steps 100 --no-pmc --no-cpu-baseline --inflight 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', '|', d['ms_per_step'], d['value'], d['roofline']['frac'])"; }
run --config cfg3
run --config cfg3 --agents 1 --force-sharded
run --config cfg3 --agents 1
run --config cfg4
run --config cfg4 --agents 2 --force-sharded
run --config cfg4 --agents 2
run --config cfg2 --force-sharded
run --config cfg2
