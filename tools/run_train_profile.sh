cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/loss
python -m pytest tests/test_loss_gpu.py tests/test_train_gpu.py -q -x 2>&1 | tail -15
python tools/bench_train.py 2>&1 | tail -4
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/proft -o t -- python $R/tools/prof_train.py > /dev/null 2> $R/gpurun_out/loss/prof.err
DB=$(find /tmp/proft -name '*.db' | head -1)
python - "$DB" > $R/gpurun_out/loss/train_kernels.txt <<'PY'
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1]); c = db.cursor()
rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
tot = sum(r[2] for r in rows)
print("total kernel us (4 steps + setup): %.0f" % (tot/1e3))
for name, calls, t, avg, pct in rows[:70]:
    print("%8d %10.1f %9.2f %6.2f%%  %s" % (calls, t/1e3, avg/1e3, pct, re.sub(r"\s+", " ", name)[:150]))
PY
head -75 $R/gpurun_out/loss/train_kernels.txt
