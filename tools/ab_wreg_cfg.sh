run() { timeout 400 python bench.py --config $CFG --no-cpu-baseline --no-pmc --inflight 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$CFG $1', d['ms_per_step'], d['value'], d['roofline']['frac'])"; }
for CFG in cfg3 cfg4; do
W2C_WREG_MINCIN=0 run none
run default93
W2C_WREG_FORM=80 run f80
W2C_WREG_MINCIN=128 run min128_93
W2C_WREG_MINCIN=128 W2C_WREG_FORM=80 run min128_80
done
