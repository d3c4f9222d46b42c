# round profile: the bench line (with PMC traffic + CPU baseline), rocprofv3 --kernel-trace --stats of the same command, digest
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-r03a}
mkdir -p $OUT
python bench.py > $OUT/bench.json 2> $OUT/bench.err
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof -o t -- python $R/bench.py --no-cpu-baseline --no-pmc --inflight 1 > $R/$OUT/prof_bench.json 2> $R/$OUT/prof.err
DB=$(find /tmp/prof -name '*.db' | head -1)
python $R/tools/rocprof_summary.py $DB --forward > $R/$OUT/kernel_trace_stats.txt 2>&1
cat $R/$OUT/bench.json
head -60 $R/$OUT/kernel_trace_stats.txt
