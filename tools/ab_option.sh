# A/B of one library option (environment-seeded at load) on the headline bench, interleaved in one call:  ab_option.sh NAME V0 V1
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 50 --warmup 5 --no-pmc --no-cpu-baseline"
for i in 1 2 3; do
  for v in $2 $3; do
    env $1=$v $B 2>/dev/null | python -c "import sys,json;d=json.loads(sys.stdin.read());print('$1=$v',d['ms_per_step'])"
  done
done
