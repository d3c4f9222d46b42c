cd $GRAFT_REPO_ROOT
for m in 0 1 2; do echo "W2C_XCD2D=$m"; W2C_XCD2D=$m W2C_LAYERS="g2 res" python tools/bench_conv.py 0 30 36 2>&1 | tail -12; done
for m in 0 1; do echo "bench W2C_XCD2D=$m"; W2C_XCD2D=$m python bench.py --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"; done
W2C_XCD2D=1 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], r['traffic'], r['traffic_over_algorithmic']); [print(k) for k in r['traffic_per_kernel'][:8]]"
