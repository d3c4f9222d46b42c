cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/loss
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/proft -o t -- python $R/tools/prof_train.py > /dev/null 2> $R/gpurun_out/loss/prof.err
DB=$(find /tmp/proft -name '*.db' | head -1)
python - "$DB" > $R/gpurun_out/loss/train_kernels.txt <<'PY'
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1]); c = db.cursor()
rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
tot = sum(r[2] for r in rows)
print("total kernel ms (all steps + MIOpen find phase): %.1f" % (tot/1e3))
for name, calls, t, avg, pct in rows[:45]:
    print("%8d %10.2f ms %9.1f us %6.2f%%  %s" % (calls, t/1e3, avg, pct, re.sub(r"\s+", " ", name)[:120]))
PY
head -50 $R/gpurun_out/loss/train_kernels.txt
