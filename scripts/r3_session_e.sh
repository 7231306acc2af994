#!/bin/bash
# round 3, session E: place on big trees, 192-byte slots (parity + launch time), per-channel counters of fast and slow scan launches
mkdir -p gpurun_out
python -m pytest tests/test_place.py::test_heavy_reads_stay_on_the_device tests/test_gpu_syn1000.py::test_place_on_the_1000_genome_tree_never_leaves_the_device "tests/test_gpu_syn1000.py::test_syn1000_10gb_index_vs_oracle_and_full_batch_properties[slotted_w48]" "tests/test_gpu_parity.py::test_overflow_path_many_leaves" -m gpu -x -q -s > gpurun_out/r3e_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r3e_tests.log
tail -6 gpurun_out/r3e_tests.log; grep "heavy reads" gpurun_out/r3e_tests.log
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-inclusive --check-reads 2000 --distinct-batches 1"
one() { name=$1; shift; echo -n "$name: "; env "$@" 2>gpurun_out/r3e_$name.err | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['kernel_ms'].items() if k in ('scan','accumulate','llh_select')}, d['check']['rows_equal'], d['config']['index_device_bytes'])"; grep stream-variance gpurun_out/r3e_$name.err | sed 's/\[stream-variance\] //'; }
one w64_a X=1 $B --stream-variance 3
one w48_a KR_SLOT_LOG2W=8 $B --stream-variance 3
one w64_b X=1 $B --stream-variance 3
one w48_b KR_SLOT_LOG2W=8 $B --stream-variance 3
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8
OUT=$PWD/gpurun_out/r3e_chan
mkdir -p $OUT
rocprofv3 --pmc TCC_EA0_RDREQ TCC_EA0_RDREQ_LEVEL TCC_EA0_RDREQ_DRAM_CREDIT_STALL TCC_EA0_WRREQ --kernel-trace --output-format json -d $OUT -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-inclusive --read-procs 1 --check-reads 2000 --distinct-batches 1 --stream-variance 4 > $OUT/bench.log 2>&1
ls -la $OUT/*/* | head; python3 scripts/chan_summary.py $OUT > gpurun_out/r3e_chan_summary.txt 2>&1; tail -40 gpurun_out/r3e_chan_summary.txt
find $OUT -name "*.json" -size +30M -delete
