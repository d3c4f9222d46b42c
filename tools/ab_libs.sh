# A/B of two builds of the library on the headline bench, interleaved in one call: tools/libA.so (default build) vs tools/libB.so
cd $GRAFT_REPO_ROOT
B="python bench.py --steps 50 --warmup 5 --no-pmc --no-cpu-baseline"
for i in 1 2 3; do
  for v in A B; do
    cp tools/lib$v.so multiagentperception_amd/lib/libw2c_hip.so
    $B 2>/dev/null | python -c "import sys,json;d=json.loads(sys.stdin.read());print('lib$v',d['ms_per_step'],d['roofline']['frac'],d['parity']['logits_rel_l2'] if 'parity' in d else '')"
  done
done
cp tools/libA.so multiagentperception_amd/lib/libw2c_hip.so
