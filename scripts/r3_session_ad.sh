#!/bin/bash
ulimit -c 0
cd "$(dirname "$0")/.."
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8
for part in c 0 1 2 3 4 x; do
  echo "== part $part"
  KR_TEST_PART=$part python3 -m pytest tests/test_gpu_parity.py -x -q -k "where_a_streams or lanes" 2>&1 | grep -E "passed|failed|At index" | cut -c1-200
done
