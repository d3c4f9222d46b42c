cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for f in 3 2; do
cd /tmp && W2C_STEM_FORM=$f rocprofv3 --kernel-trace --stats -d /tmp/prof$f -o t -- python $R/bench.py --no-cpu-baseline --no-pmc > /tmp/b$f.json 2>/dev/null
DB=$(find /tmp/prof$f -name '*.db' | head -1)
echo "FORM $f: $(cut -c80-140 /tmp/b$f.json)"
python $R/tools/rocprof_summary.py $DB | grep -i "stem"
done
