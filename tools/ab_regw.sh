run() { timeout 300 python bench.py --no-cpu-baseline --no-pmc --inflight 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['value'], d['roofline']['kernel_ms_per_step'])"; }
for i in 1 2 3; do
W2C_REGW_FORM=1 run staged
W2C_REGW_FORM=2 run direct
done
