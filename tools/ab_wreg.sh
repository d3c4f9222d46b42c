run() { timeout 300 python bench.py --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['value'], d['roofline']['kernel_ms_per_step'])"; }
for i in 1 2; do
W2C_WREG_MINCIN=0 run none
run default93
W2C_WREG_FORM=81 run f81
W2C_WREG_MINCIN=128 run min128_93
W2C_WREG_MINCIN=128 W2C_WREG_FORM=80 run min128_80
W2C_WREG_FORM=80 run f80
done
