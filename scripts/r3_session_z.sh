#!/bin/bash
# Round 3, session Z: select kernel with four reads in flight (unrolled rotation), dedup kernel with one sc1 16-byte probe
ulimit -c 0
cd "$(dirname "$0")/.."
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out
python3 -m pytest tests/test_gpu_parity.py -x -q -k "golden or report_modes or large_batch or device_brent or where_a_streams or lanes or long_reads" 2>&1 | tail -3
python3 -m pytest tests/test_gpu_syn1000.py -x -q -k "slotted" 2>&1 | tail -3
rm -rf /tmp/pytest-of-* /tmp/krepp_*
trace() {
  OUT=$PWD/gpurun_out/r3z_$1; rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-inclusive --read-procs 1 --distinct-batches 1 > $OUT/bench.log 2>&1
  echo "== $1"; python3 scripts/kstats.py $OUT | grep -E "select|dedup_kernel|llh_kernel" | cut -c1-200
  rm -rf /tmp/krepp_bench_*
}
trace new
