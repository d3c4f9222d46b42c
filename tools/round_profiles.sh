#!/bin/bash
# End-of-stage evidence in one GPU call: full bench line (PMC traffic + CPU baseline), kernel trace, SQ counters, rank shapes, output kernels.
# usage: tools/round_profiles.sh <tag>      -> gpurun_out/<tag>_*
tag=$1
mkdir -p gpurun_out
python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
tools/prof_forward.sh ${tag}
export TMPDIR=/tmp
tools/sq_counters.sh ${tag} python $PWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pmc --no-graph --inflight 1
bash tools/run_rank_shapes2.sh > gpurun_out/${tag}_rank_shapes_raw.txt 2>&1
bash tools/run_rank_shapes.sh >> gpurun_out/${tag}_rank_shapes_raw.txt 2>&1
python tools/bench_upsample.py > gpurun_out/${tag}_output_kernels.txt 2>&1
python tools/chain_stamps.py > gpurun_out/${tag}_chain_stamps.txt 2>&1
for cfg in "--config cfg2 --mode activated" "--config cfg5 --bf16"; do
  python bench.py $cfg --no-cpu-baseline --no-pmc --inflight 1 2>/dev/null | tail -1 >> gpurun_out/${tag}_bench_other.jsonl
done
