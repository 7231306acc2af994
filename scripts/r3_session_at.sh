#!/bin/bash
# round 3, session AT: the parity sweeps beyond the fixed tests, on the final code
ulimit -c 0
mkdir -p gpurun_out
for s in sweep_configs sweep_libs sweep_place sweep_seek; do
  echo "== $s"
  timeout 600 python3 scripts/$s.py 2>&1 | tail -4 | cut -c1-250
done
echo "== fuzz_reads 12"; timeout 600 python3 scripts/fuzz_reads.py 12 2>&1 | tail -2
echo "== fuzz_long 40"; timeout 900 python3 scripts/fuzz_long.py 40 2>&1 | tail -2
