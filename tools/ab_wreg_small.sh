cd $GRAFT_REPO_ROOT
run() { python bench.py "$@" --steps 40 --no-pmc --no-cpu-baseline --inflight 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$W2C_WREG_SMALL', '$*', '|', d['ms_per_step'])"; }
for i in 1 2 3; do for f in 0 300; do export W2C_WREG_SMALL=$f
run --config cfg3 --agents 1
run --config cfg4 --agents 2
run
done; done
