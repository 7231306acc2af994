#!/bin/bash
# Sample clocks and power while a command runs: scripts/smi_sample.sh <out.txt> <command...>   (experiment: does the scan kernel's
# two-mode launch time follow a clock or power state?)
OUT=$1; shift
( while true; do date +%s.%N; /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|socclk|Power" ; sleep 0.2; done ) > $OUT 2>&1 &
SPID=$!
"$@"
kill $SPID
