#!/bin/bash
ulimit -c 0
cd "$(dirname "$0")/.."
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8
echo "== all buffers poisoned"
KR_DEBUG_POISON=all python3 -m pytest tests/test_gpu_parity.py -x -q -k "golden or lanes" 2>&1 | grep -E "passed|failed|At index|Error" | cut -c1-200
for k in $(seq 1 45); do
  r=$(KR_DEBUG_POISON=$k python3 -m pytest tests/test_gpu_parity.py -x -q -k "hits_accumulators_rows_match" 2>&1 | grep -E "passed|failed" | cut -c1-60)
  echo "buffer $k: $r"
done
