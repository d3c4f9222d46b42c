# repeat the headline bench (and the one-rank sharded cfg 3 share) in fresh processes and print ms_per_step + the host enqueue gaps:
# hunting the rare ~2.97 ms/step state seen twice this round (gpurun_out/r04_e_rank_shapes_raw.txt line 1)
cd $GRAFT_REPO_ROOT
n=${1:-12}
for i in $(seq 1 $n); do
  for cfgargs in "" "--config cfg3 --agents 1 --force-sharded"; do
    python bench.py $cfgargs --steps 20 --no-pmc --no-cpu-baseline --inflight 1 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$i', '$cfgargs' or 'cfg2', d['ms_per_step'], d['host_enqueue'], d['roofline'].get('kernel_ms_per_step'))"
  done
done
