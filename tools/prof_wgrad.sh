cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for L in "l1 64" "l4 512"; do
cd /tmp && rm -rf /tmp/pw && W2C_LAYERS="$L" rocprofv3 --kernel-trace --stats -d /tmp/pw -o t -- python $R/tools/bench_wgrad.py > /dev/null 2>&1
DB=$(find /tmp/pw -name '*.db' | head -1)
echo "== $L"; python $R/tools/rocprof_summary.py $DB | grep -i "wgrad" 
done
