#!/bin/bash
# round 5: reproduce the SIGSEGV of the SRMS graph-replay test (GPUTEST_r04 rc=139) with full logs
out=gpurun_out/r05_a; mkdir -p $out
export PYTHONFAULTHANDLER=1
for i in $(seq 1 ${1:-10}); do
  timeout 600 python -m pytest tests/test_srms.py -x -q -m gpu -p no:cacheprovider > $out/srms_$i.log 2>&1
  rc=$?; echo "iter $i rc=$rc" >> $out/summary.txt
  if [ $rc -ne 0 ]; then cp $out/srms_$i.log $out/srms_FAIL_$i.log; fi
  tail -1 $out/srms_$i.log >> $out/summary.txt
done
# full suite under rocgdb for a native backtrace
cat > /tmp/gdbcmds <<'G'
set pagination off
set confirm off
handle SIGPIPE nostop noprint pass
handle SIG32 nostop noprint pass
handle SIG33 nostop noprint pass
handle SIG34 nostop noprint pass
handle SIGUSR1 nostop noprint pass
run
bt 40
info threads
thread apply all bt 30
G
timeout 1500 rocgdb -batch -x /tmp/gdbcmds --args python -m pytest tests/ -x -q -m gpu -p no:cacheprovider --durations=25 > $out/full_gdb.log 2>&1
echo "full gdb rc=$?" >> $out/summary.txt
tail -5 $out/full_gdb.log >> $out/summary.txt
cat $out/summary.txt
