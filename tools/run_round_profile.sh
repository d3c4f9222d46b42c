cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02d
python -m pytest tests -m gpu -q -x > gpurun_out/r02d/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02d/pytest.log
python bench.py > gpurun_out/r02d/bench.json 2> gpurun_out/r02d/bench.err
python bench.py --config cfg5 --no-cpu-baseline > gpurun_out/r02d/bench_cfg5.json 2> gpurun_out/r02d/bench_cfg5.err
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof -o t -- python $R/bench.py --no-cpu-baseline --no-pmc > $R/gpurun_out/r02d/prof_bench.json 2> $R/gpurun_out/r02d/prof.err
DB=$(find /tmp/prof -name '*.db' | head -1)
python $R/tools/rocprof_summary.py $DB --forward > $R/gpurun_out/r02d/kernel_trace_stats.txt 2>&1
tail -3 $R/gpurun_out/r02d/pytest.log
cat $R/gpurun_out/r02d/bench.json $R/gpurun_out/r02d/bench_cfg5.json
