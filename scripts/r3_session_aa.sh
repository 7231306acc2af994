#!/bin/bash
# Round 3, session AA: the lanes test with the 16-byte dedup probe and with the two 8-byte atomic loads
ulimit -c 0
cd "$(dirname "$0")/.."
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out
echo "== main (16-byte probe), lanes test alone"
python3 -m pytest tests/test_gpu_parity.py -x -q -k "lanes" 2>&1 | tail -40 | cut -c1-300
cp krepp_amd/lib/libkrepp_amd.so /tmp/main_lib.so
cp krepp_amd/lib/variants/probe8/libkrepp_amd.so krepp_amd/lib/libkrepp_amd.so
echo "== probe8, lanes test alone"
python3 -m pytest tests/test_gpu_parity.py -x -q -k "lanes" 2>&1 | tail -5 | cut -c1-300
cp /tmp/main_lib.so krepp_amd/lib/libkrepp_amd.so
