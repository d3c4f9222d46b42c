# per-kernel trace of one rank's share of cfg 3 (1 agent x B=8 x 512^2) as a plain forward: tools/r06/prof_rank.sh [extra bench args]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/rank
cd /tmp && rm -rf /tmp/profr && rocprofv3 --kernel-trace --stats -d /tmp/profr -o t -- python $R/bench.py --config cfg3 --agents 1 --no-cpu-baseline --no-pmc --inflight 1 "$@" > $R/gpurun_out/rank/prof_bench.json 2> $R/gpurun_out/rank/prof.err
DB=$(find /tmp/profr -name '*.db' | head -1)
python $R/tools/rocprof_summary.py $DB --forward > $R/gpurun_out/rank/kernel_trace_stats.txt 2>&1
