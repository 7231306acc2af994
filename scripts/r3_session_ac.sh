#!/bin/bash
ulimit -c 0
cd "$(dirname "$0")/.."
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out
echo "== where_a_streams + lanes"
python3 -m pytest tests/test_gpu_parity.py -x -q -vv -k "where_a_streams or lanes" 2>&1 | grep -v "^$" | tail -60 | cut -c1-400
echo "== everything before the new test + lanes (new test deselected)"
python3 -m pytest tests/test_gpu_parity.py -x -q -k "not where_a_streams" 2>&1 | tail -5 | cut -c1-300
