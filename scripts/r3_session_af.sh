#!/bin/bash
# Round 3, session AF: select kernel with the packed word fetched with the record; 32 lanes per read
ulimit -c 0
cd "$(dirname "$0")/.."
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out
python3 -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3 | cut -c1-200
rm -rf /tmp/pytest-of-* /tmp/krepp_*
trace() {
  OUT=$PWD/gpurun_out/r3af_$1; rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-inclusive --read-procs 1 --distinct-batches 1 > $OUT/bench.log 2>&1
  echo "== $1"; python3 scripts/kstats.py $OUT | grep -E "select|dedup_kernel" | cut -c1-200
  grep -h '"metric"' $OUT/bench.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('   rows_equal', d['check']['rows_equal'], 'llh_select', round(d['kernel_ms']['llh_select'],2))"
  rm -rf /tmp/krepp_bench_*
}
trace main
cp krepp_amd/lib/libkrepp_amd.so /tmp/main_lib.so
for v in w0late gl32; do
  cp krepp_amd/lib/variants/$v/libkrepp_amd.so krepp_amd/lib/libkrepp_amd.so
  trace $v
done
cp /tmp/main_lib.so krepp_amd/lib/libkrepp_amd.so
