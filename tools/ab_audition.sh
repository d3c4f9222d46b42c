cd $GRAFT_REPO_ROOT
for q in 5 4 6; do for a in 1 3; do
GPU_MAX_HW_QUEUES=$q W2C_GRAPH_AUDITION=$a python bench.py --steps 40 --warmup 5 --no-pmc --no-cpu-baseline --no-retry 2>/dev/null | python -c "import sys,json;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('Q=$q audition=$a',d['ms_per_step'], d['config'].get('graph_audition_ms'), d['forwards_in_flight']['ms_per_forward'], d['evaluator_path']['ms_per_step'])"
done; done
