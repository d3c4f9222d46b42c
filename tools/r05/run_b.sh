#!/bin/bash
out=gpurun_out/r05_b; mkdir -p $out
timeout 900 python tools/r05/gc_in_capture.py > $out/gc_in_capture.txt 2>&1
cat > /tmp/gdbcmds <<'G'
set pagination off
set confirm off
handle SIGPIPE nostop noprint pass
run
bt 40
G
timeout 300 rocgdb -batch -x /tmp/gdbcmds --args python tools/r05/gc_in_capture.py gc_old_graph_in_capture > $out/gdb_gc_in_capture.txt 2>&1
timeout 300 python tools/r05/fr_probe.py > $out/fr_probe.txt 2>&1
cat $out/gc_in_capture.txt; tail -60 $out/gdb_gc_in_capture.txt; cat $out/fr_probe.txt
