#!/bin/bash
# rocprofv3 --kernel-trace --stats of the default bench command (cfg 2, graph replay) -> gpurun_out/<tag>_kernel_trace_stats.txt
# usage: tools/prof_forward.sh <tag> [extra bench args]
tag=$1; shift
export TMPDIR=/tmp
out=$PWD/gpurun_out
mkdir -p $out
d=$(mktemp -d /tmp/w2c_prof_XXXX)
( cd /tmp && rocprofv3 --kernel-trace --stats -d $d -o t -- python $OLDPWD/bench.py --no-cpu-baseline --no-pmc --inflight 1 "$@" > $out/${tag}_prof_bench.json 2> $out/${tag}_prof_bench.err )
db=$(find $d -name "*.db" | head -1)
python tools/rocprof_summary.py $db --forward > $out/${tag}_kernel_trace_stats.txt
rm -rf $d
