cd $GRAFT_REPO_ROOT
bash tools/run_round_profile.sh r03e > /dev/null 2>&1
bash tools/run_sq_counters.sh > /dev/null 2>&1
python bench.py --config cfg3 --no-cpu-baseline --no-pmc > gpurun_out/r03e/bench_cfg3_1gpu.json 2>/dev/null
python bench.py --config cfg4 --no-cpu-baseline --no-pmc > gpurun_out/r03e/bench_cfg4_1gpu.json 2>/dev/null
python bench.py --config cfg5 --bf16 --no-cpu-baseline --no-pmc > gpurun_out/r03e/bench_cfg5_bf16.json 2>/dev/null
python bench.py --mode activated --no-cpu-baseline --no-pmc > gpurun_out/r03e/bench_cfg2_activated.json 2>/dev/null
python -c "
import json,glob
for f in sorted(glob.glob('gpurun_out/r03e/bench*.json')):
    try:
        d=json.load(open(f)); r=d['roofline']; print(f.split('/')[-1], d['value'], d['ms_per_step'], r['frac'], r.get('frac_by_sum_of_durations'), d.get('parity',{}).get('logits_rel_l2'))
    except Exception as e: print(f, 'ERR', e)
"
tail -4 gpurun_out/r03e/kernel_trace_stats.txt
head -20 gpurun_out/sq/sq_counters.txt | cut -c1-150
