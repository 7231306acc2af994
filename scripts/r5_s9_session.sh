#!/bin/bash
# round 5, session 9: the whole GPU suite and smoke() at HEAD
ulimit -c 0
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q --durations=6 > gpurun_out/r5_s9_tests.txt 2>&1
grep -n "passed\|failed\|error" gpurun_out/r5_s9_tests.txt | head; tail -9 gpurun_out/r5_s9_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5_s9_smoke.txt 2>&1; tail -2 gpurun_out/r5_s9_smoke.txt
