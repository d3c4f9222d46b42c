run() { timeout 300 python bench.py --no-cpu-baseline --no-pmc --inflight 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['value'])"; }
for i in 1 2 3; do W2C_HEADS_AFTER_JOIN=1 run join; run chain; done
