# Round-3 GPU sessions: the `gpurun` command scripts of that round, one after the other (formerly scripts/r3_session_<id>.sh).
# A record of what was run, not something to run: each section was the whole script of one session; results went to gpurun_out/r3<id>*
# and, where they were kept, to profiles/round3_*.  Kept in one file so that scripts/ lists tools, not logs.

#### session a ############################################################
#!/bin/bash
# round 3, first GPU session: the new tests first, then the whole GPU suite, then the default bench
mkdir -p gpurun_out
python -m pytest tests/test_gpu_place_k27.py tests/test_gpu_rccl_cli.py tests/test_gpu_bench.py -m gpu -x -q > gpurun_out/r3a_new_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r3a_new_tests.log
tail -15 gpurun_out/r3a_new_tests.log
python -m pytest tests -m gpu -q --durations=15 > gpurun_out/r3a_pytest_gpu.log 2>&1; echo "rc=$?" >> gpurun_out/r3a_pytest_gpu.log
tail -30 gpurun_out/r3a_pytest_gpu.log
python bench.py --steps 10 --warmup 2 > gpurun_out/r3a_bench.json 2> gpurun_out/r3a_bench.err; tail -c 1500 gpurun_out/r3a_bench.json

#### session b ############################################################
#!/bin/bash
# round 3, session B: the pipelined scan -- parity first, then launch times of the variants at several occupancies
mkdir -p gpurun_out
python -m pytest tests/test_gpu_place_k27.py tests/test_gpu_rccl_cli.py "tests/test_gpu_syn1000.py::test_syn1000_10gb_index_vs_oracle_and_full_batch_properties[slotted_w64]" tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r3b_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r3b_tests.log
tail -8 gpurun_out/r3b_tests.log
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-host-inclusive --check-reads 2000 --distinct-batches 1"
one() { # name, env...
  name=$1; shift
  echo -n "$name: "
  env "$@" $B 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), {k:round(v,2) for k,v in d['kernel_ms'].items() if k in ('scan','accumulate','llh_select')}, d['check']['rows_equal'])"
}
cp krepp_amd/lib/libkrepp_amd.so /tmp/main_lib.so
one old_w4 KR_SCAN_PIPE=0
one old_b3 KR_SCAN_PIPE=0 KR_DEBUG_SCAN_BLOCKS_PER_CU=3
one old_b2 KR_SCAN_PIPE=0 KR_DEBUG_SCAN_BLOCKS_PER_CU=2
for v in d1w4 d2w4 d2w3 d3w3 d4w2; do
  cp krepp_amd/lib/variants/$v/libkrepp_amd.so krepp_amd/lib/libkrepp_amd.so
  one $v X=1
  one ${v}_b3 KR_DEBUG_SCAN_BLOCKS_PER_CU=3
  one ${v}_b2 KR_DEBUG_SCAN_BLOCKS_PER_CU=2
done
cp /tmp/main_lib.so krepp_amd/lib/libkrepp_amd.so

#### session c ############################################################
#!/bin/bash
# round 3, session C: scan of batch i+1 beside accumulate / likelihood of batch i (two kernel chains)
mkdir -p gpurun_out
B="python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-host-inclusive --check-reads 2000 --distinct-batches 1"
one() { # name, args..., env via leading VAR=VAL words
  name=$1; shift
  echo -n "$name: "
  env "$@" 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['kernel_ms'].items() if k in ('scan','accumulate','llh_select')}, d['check']['rows_equal'])"
}
one serial X=1 $B
one serial_2streams X=1 $B --pipeline-streams 2
one acc8 KR_DEBUG_ACC_WAVES=8 $B
one acc12 KR_DEBUG_ACC_WAVES=12 $B
one acc16 KR_DEBUG_ACC_WAVES=16 $B
one ovl_d2_s2_a8 KR_OVERLAP=1 $B --pipeline-streams 2
one ovl_d2_s2_a12 KR_OVERLAP=1 KR_OVERLAP_ACC_WAVES=12 $B --pipeline-streams 2
one ovl_d2_s2_a6 KR_OVERLAP=1 KR_OVERLAP_ACC_WAVES=6 $B --pipeline-streams 2
one ovl_d1_s2_a12 KR_OVERLAP=1 KR_OVERLAP_SCAN_D1=1 KR_OVERLAP_ACC_WAVES=12 $B --pipeline-streams 2
one ovl_d1_s2_a8 KR_OVERLAP=1 KR_OVERLAP_SCAN_D1=1 KR_OVERLAP_ACC_WAVES=8 $B --pipeline-streams 2
one ovl_d1_s3_a4 KR_OVERLAP=1 KR_OVERLAP_SCAN_D1=1 KR_OVERLAP_SCAN_BLOCKS=3 KR_OVERLAP_ACC_WAVES=4 $B --pipeline-streams 2
one ovl_d2_s3_a8 KR_OVERLAP=1 KR_OVERLAP_SCAN_BLOCKS=3 KR_OVERLAP_ACC_WAVES=8 $B --pipeline-streams 2
one ovl_d1_s4_a24 KR_OVERLAP=1 KR_OVERLAP_SCAN_D1=1 KR_OVERLAP_SCAN_BLOCKS=4 KR_OVERLAP_ACC_WAVES=24 $B --pipeline-streams 2
one ovl_d2_s2_a8_3streams KR_OVERLAP=1 $B --pipeline-streams 3

#### session d ############################################################
#!/bin/bash
# round 3, session D: place tests (heavy reads on the device), counter list, stream-to-stream variance of the pipelined scan
mkdir -p gpurun_out
python -m pytest tests/test_place.py tests/test_gpu_place_k27.py tests/test_gpu_rccl_cli.py tests/test_gpu_syn1000.py::test_place_on_the_1000_genome_tree_never_leaves_the_device -m gpu -x -q -s > gpurun_out/r3d_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r3d_tests.log
tail -8 gpurun_out/r3d_tests.log; grep "heavy reads" gpurun_out/r3d_tests.log
cd /tmp && export TMPDIR=/tmp && rocprofv3 -L > $GRAFT_REPO_ROOT/gpurun_out/r3d_counters.txt 2>&1; cd $GRAFT_REPO_ROOT
grep -c . gpurun_out/r3d_counters.txt; grep -i "TCC_EA0_RDREQ\b\|TCC_EA0_WRREQ\b\|TCC_REQ\b\|TCP_TCC_READ_REQ\b\|MALL\|TCC_EA0_RD_UNCACHED" gpurun_out/r3d_counters.txt | head -20
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-host-inclusive --check-reads 2000 --distinct-batches 1 --stream-variance 8"
for i in 1 2; do $B 2> gpurun_out/r3d_var$i.err | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), {k:round(v,2) for k,v in d['kernel_ms'].items() if k in ('scan','accumulate','llh_select')})"; grep stream-variance gpurun_out/r3d_var$i.err | sed 's/\[stream-variance\] //'; done

#### session e ############################################################
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

#### session f ############################################################
#!/bin/bash
# round 3, session F: what the slow scan launches are (address-translation counters), item chunk size, likelihood stage beside the next scan
mkdir -p gpurun_out
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-inclusive --check-reads 2000 --distinct-batches 1"
one() { name=$1; shift; echo -n "$name: "; env "$@" 2>gpurun_out/r3f_$name.err | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['kernel_ms'].items() if k in ('scan','accumulate','llh_select')}, d['check']['rows_equal'])"; grep stream-variance gpurun_out/r3f_$name.err | sed 's/\[stream-variance\] //'; }
# --- overlap mode 2: dedup .. select of batch i beside the scan of batch i+1
one serial X=1 $B
one ovl2_d1_s3 KR_OVERLAP=2 KR_OVERLAP_SCAN_D1=1 KR_OVERLAP_SCAN_BLOCKS=3 $B --pipeline-streams 2
one ovl2_d1_s4 KR_OVERLAP=2 KR_OVERLAP_SCAN_D1=1 KR_OVERLAP_SCAN_BLOCKS=4 $B --pipeline-streams 2
one ovl2_d2_s2 KR_OVERLAP=2 KR_OVERLAP_SCAN_BLOCKS=2 $B --pipeline-streams 2
one ovl2_d2_s3 KR_OVERLAP=2 KR_OVERLAP_SCAN_BLOCKS=3 $B --pipeline-streams 2
# --- item chunk size against the stream-to-stream levels of the scan
cp krepp_amd/lib/libkrepp_amd.so /tmp/main_lib.so
for v in chunk512 chunk256; do
  cp krepp_amd/lib/variants/$v/libkrepp_amd.so krepp_amd/lib/libkrepp_amd.so
  one ${v}_a X=1 $B --stream-variance 4
  one ${v}_b X=1 $B --stream-variance 4
done
cp /tmp/main_lib.so krepp_amd/lib/libkrepp_amd.so
one chunk2048_a X=1 $B --stream-variance 4
# --- address translation counters per scan launch (fast and slow streams in one process)
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8
OUT=$PWD/gpurun_out/r3f_tlb
mkdir -p $OUT
rocprofv3 --pmc TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_THRASHING_STALL_sum --kernel-trace --output-format csv -d $OUT -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-inclusive --read-procs 1 --check-reads 2000 --distinct-batches 1 --stream-variance 5 > $OUT/bench.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/r3f_tlb/**/*counter_collection.csv', recursive=True)
print(f)
rows = collections.OrderedDict()
for r in csv.DictReader(open(f[0])):
    if 'kr_scan' not in r['Kernel_Name']: continue
    k = int(r['Dispatch_Id'])
    rows.setdefault(k, {'dur': (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6})[r['Counter_Name']] = float(r['Counter_Value'])
for k, v in rows.items():
    print(k, {a: (round(b, 2) if a == 'dur' else f'{b:.4g}') for a, b in v.items()})
PY

#### session g ############################################################
#!/bin/bash
# round 3, session G: is the scan's launch-time level a property of the hardware queue or of the buffers?  place throughput on the 1000-genome tree; 250-bp reads
mkdir -p gpurun_out
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-host-inclusive --check-reads 2000 --distinct-batches 1"
one() { name=$1; shift; echo -n "$name: "; env "$@" 2>gpurun_out/r3g_$name.err | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['kernel_ms'].items() if k in ('scan','accumulate','llh_select')}, d['check']['rows_equal'])"; grep stream-variance gpurun_out/r3g_$name.err | sed 's/\[stream-variance\] //'; }
one q0_a KR_DEBUG_EXTRA_STREAMS=0 $B --stream-variance 5
one q0_b KR_DEBUG_EXTRA_STREAMS=0 $B --stream-variance 5
one q1_a KR_DEBUG_EXTRA_STREAMS=1 $B --stream-variance 5
one q1_b KR_DEBUG_EXTRA_STREAMS=1 $B --stream-variance 5
one q3_a KR_DEBUG_EXTRA_STREAMS=3 $B --stream-variance 5
one hwq1 GPU_MAX_HW_QUEUES=1 $B --stream-variance 5
python scripts/time_place_big.py 400000 2>&1 | tail -4
one len250 X=1 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-host-inclusive --check-reads 2000 --distinct-batches 1 --read-len 250 --reads-per-step 4000000

#### session i ############################################################
#!/bin/bash
# round 3, session I: the whole GPU suite on the current code, the profile passes behind profiles/round3_a_*, place timing on the 1000-genome tree
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --durations=8 > gpurun_out/r3i_pytest_gpu.log 2>&1; echo "rc=$?" >> gpurun_out/r3i_pytest_gpu.log
tail -14 gpurun_out/r3i_pytest_gpu.log
bash scripts/profile.sh r3a > gpurun_out/prof_r3a.log 2>&1
tail -5 gpurun_out/prof_r3a.log
python bench.py --steps 10 --warmup 2 > gpurun_out/r3i_bench.json 2> gpurun_out/r3i_bench.err
python scripts/traffic.py gpurun_out/prof_r3a gpurun_out/r3i_bench.json gpurun_out/traffic_r3a.json
KR_PLACE_TIMING=1 python scripts/time_place_big.py 400000 > gpurun_out/r3i_place.log 2>&1; grep -v "^\[place" gpurun_out/r3i_place.log | tail -4; grep "place/device\|\[place\]" gpurun_out/r3i_place.log | tail -12

#### session j ############################################################
#!/bin/bash
# round 3, session J: the two-segment accumulate instantiation (parity, 250-bp rate); place with compacted candidates; what the device phase of place spends its time on
ulimit -c 0
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_place.py tests/test_gpu_place_k27.py tests/test_gpu_syn1000.py::test_place_on_the_1000_genome_tree_never_leaves_the_device -m gpu -x -q > gpurun_out/r3j_tests.log 2>&1; rc=$?; echo "rc=$rc" >> gpurun_out/r3j_tests.log; tail -5 gpurun_out/r3j_tests.log
if [ $rc -ne 0 ]; then grep -n "^E " gpurun_out/r3j_tests.log | head -20; exit 0; fi
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-host-inclusive --check-reads 4000 --distinct-batches 1"
one() { name=$1; shift; echo -n "$name: "; env "$@" 2>gpurun_out/r3j_$name.err | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['kernel_ms'].items() if k in ('scan','accumulate','llh_select')}, d['check'])"; rm -rf /tmp/krepp_bench_*; }
one len250_small X=1 $B --read-len 250 --reads-per-step 200000
one len250 X=1 $B --read-len 250 --reads-per-step 4000000
one len200 X=1 $B --read-len 200 --reads-per-step 4000000
one len300 X=1 $B --read-len 300 --reads-per-step 4000000
one len150 X=1 $B
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8
OUT=$PWD/gpurun_out/r3j_place_trace
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 scripts/time_place_big.py 400000 > $OUT/run.log 2>&1
tail -3 $OUT/run.log
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/r3j_place_trace/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'kr_' in r['Name']:
        print(r['Name'][:80], r['Calls'], 'avg_ms', round(float(r['AverageNs'])/1e6, 3), 'max_ms', round(float(r['MaxNs'])/1e6, 3))
PY
find $OUT -name "*.csv" -size +5M -delete

#### session k ############################################################
#!/bin/bash
# round 3, session K: place with precomputed (leaf, ancestor) weights; event region of the two-segment accumulate instantiation; the 10,000-genome workload
ulimit -c 0
mkdir -p gpurun_out
python -m pytest tests/test_place.py tests/test_gpu_place_k27.py tests/test_gpu_syn1000.py::test_place_on_the_1000_genome_tree_never_leaves_the_device -m gpu -x -q > gpurun_out/r3k_tests.log 2>&1; rc=$?; echo "rc=$rc" >> gpurun_out/r3k_tests.log; tail -4 gpurun_out/r3k_tests.log
if [ $rc -ne 0 ]; then grep -n "^E " gpurun_out/r3k_tests.log | head -20; fi
KR_PLACE_TIMING=1 python scripts/time_place_big.py 400000 > gpurun_out/r3k_place.log 2>&1; grep -v "^\[place" gpurun_out/r3k_place.log | tail -3; grep "place/device" gpurun_out/r3k_place.log | tail -6
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-host-inclusive --check-reads 4000 --distinct-batches 1"
one() { name=$1; shift; echo -n "$name: "; env "$@" 2>gpurun_out/r3k_$name.err | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['kernel_ms'].items() if k in ('scan','accumulate','llh_select')}, d['check']['rows_equal'])"; rm -rf /tmp/krepp_bench_*; }
one len250_ev4 X=1 $B --read-len 250 --reads-per-step 4000000
cp krepp_amd/lib/libkrepp_amd.so /tmp/main_lib.so
for v in ev3 ev6; do cp krepp_amd/lib/variants/$v/libkrepp_amd.so krepp_amd/lib/libkrepp_amd.so; one len250_$v X=1 $B --read-len 250 --reads-per-step 4000000; done
cp /tmp/main_lib.so krepp_amd/lib/libkrepp_amd.so
one syn10000 X=1 python bench.py --workload syn10000 --steps 6 --warmup 2 --no-cpu-baseline --no-host-inclusive --check-reads 4000 --distinct-batches 1

#### session l ############################################################
#!/bin/bash
# round 3, session L: long sequences across waves (parity, rate), the whole GPU suite, 250-bp and 10,000-genome rates after the fixes
ulimit -c 0
mkdir -p gpurun_out
python -m pytest tests/test_gpu_long_sequences.py -m gpu -x -q > gpurun_out/r3l_tiles.log 2>&1; rc=$?; echo "rc=$rc" >> gpurun_out/r3l_tiles.log; tail -4 gpurun_out/r3l_tiles.log
if [ $rc -ne 0 ]; then grep -n "^E " gpurun_out/r3l_tiles.log | head -30; fi
python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_long_sequences.py > gpurun_out/r3l_pytest_gpu.log 2>&1; echo "rc=$?" >> gpurun_out/r3l_pytest_gpu.log; tail -6 gpurun_out/r3l_pytest_gpu.log
if [ $rc -eq 0 ]; then python scripts/time_contigs.py 400000 8 2>&1 | tail -4; python scripts/time_contigs.py 5000 2000 2>&1 | tail -4; fi
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-host-inclusive --check-reads 4000 --distinct-batches 1"
one() { name=$1; shift; echo -n "$name: "; env "$@" 2>gpurun_out/r3l_$name.err | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['kernel_ms'].items() if k in ('scan','accumulate','llh_select')}, d['check']['rows_equal'])"; rm -rf /tmp/krepp_bench_*; }
one syn10000 X=1 python bench.py --workload syn10000 --steps 6 --warmup 2 --no-cpu-baseline --no-host-inclusive --check-reads 4000 --distinct-batches 1
one len250 X=1 $B --read-len 250 --reads-per-step 4000000
one len150 X=1 $B

#### session m ############################################################
#!/bin/bash
# round 3, session M: kernel trace of place on the 1000-genome tree (after the weights / compaction changes)
ulimit -c 0
mkdir -p gpurun_out
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8
OUT=$PWD/gpurun_out/r3m_place_trace
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 scripts/time_place_big.py 400000 > $OUT/run.log 2>&1
grep -v "^\[\|^W\|^E\|^I" $OUT/run.log | tail -4
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/r3m_place_trace/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'kr_' in r['Name']:
        print(r['Name'][:80], r['Calls'], 'avg_ms', round(float(r['AverageNs'])/1e6, 3), 'max_ms', round(float(r['MaxNs'])/1e6, 3))
PY
find $OUT -name "*.csv" -size +5M -delete
KR_PLACE_TIMING=1 python scripts/time_place_big.py 400000 > gpurun_out/r3m_place.log 2>&1; grep "place/device" gpurun_out/r3m_place.log | tail -4
python scripts/time_place.py 400000 2>&1 | tail -12

#### session n ############################################################
#!/bin/bash
# round 3, session N: place on the 25-reference index after the heuristic (weights precomputed only on deep trees): phases and kernels
ulimit -c 0
mkdir -p gpurun_out
KR_PLACE_TIMING=1 python scripts/time_place.py 400000 > gpurun_out/r3n_place_toy.log 2>&1
grep -v "^\[place" gpurun_out/r3n_place_toy.log | tail -14; grep "place/device" gpurun_out/r3n_place_toy.log | sed -n '4,9p'
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8
OUT=$PWD/gpurun_out/r3n_place_trace
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 scripts/time_place.py 400000 > $OUT/run.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/r3n_place_trace/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'kr_' in r['Name'] and float(r['AverageNs']) > 2e4:
        print(r['Name'][:80], r['Calls'], 'avg_ms', round(float(r['AverageNs'])/1e6, 3), 'max_ms', round(float(r['MaxNs'])/1e6, 3))
PY
find $OUT -name "*.csv" -size +5M -delete
python scripts/time_place_big.py 400000 2>&1 | tail -3
python -m pytest tests/test_place.py tests/test_gpu_place_k27.py -m gpu -x -q 2>&1 | tail -2

#### session o ############################################################
#!/bin/bash
# round 3, session O: tile merge spread over several waves per sequence (parity, rate)
ulimit -c 0
mkdir -p gpurun_out
python -m pytest tests/test_gpu_long_sequences.py tests/test_gpu_parity.py::test_long_reads_and_ragged_batches tests/test_gpu_parity.py::test_cli_contig_queries_multiline_fasta -m gpu -x -q 2>&1 | tail -3
python scripts/time_contigs.py 400000 8 2>&1 | tail -3
python scripts/time_contigs.py 400000 1 2>&1 | tail -3
python scripts/time_contigs.py 5000 2000 2>&1 | tail -3
python scripts/time_contigs.py 50000 200 2>&1 | tail -3

#### session p ############################################################
#!/bin/bash
# round 3, session P: does the third accumulate launch cost the 150-bp workload anything?  (accumulate 21.1 ms before it, 22.1 after)
ulimit -c 0
mkdir -p gpurun_out
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-inclusive --check-reads 2000 --distinct-batches 1"
one() { name=$1; shift; echo -n "$name: "; env "$@" 2>gpurun_out/r3p_$name.err | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['kernel_ms'].items() if k in ('scan','accumulate','llh_select')}, d['check']['rows_equal'])"; rm -rf /tmp/krepp_bench_*; }
one lean2_on_a X=1 $B
one lean2_off_a KR_DEBUG_NO_LEAN2=1 $B
one lean2_on_b X=1 $B
one lean2_off_b KR_DEBUG_NO_LEAN2=1 $B

#### session q ############################################################
#!/bin/bash
# round 3, session Q: accumulate kernel before / after the two-segment instantiation (same box)
ulimit -c 0
mkdir -p gpurun_out
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-inclusive --check-reads 2000 --distinct-batches 1"
one() { name=$1; shift; echo -n "$name: "; env "$@" 2>gpurun_out/r3q_$name.err | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['kernel_ms'].items() if k in ('scan','accumulate','llh_select')}, d['check']['rows_equal'])"; rm -rf /tmp/krepp_bench_*; }
cp krepp_amd/lib/libkrepp_amd.so /tmp/main_lib.so
one head_a X=1 $B
for v in old40d8 mid95f4; do cp krepp_amd/lib/variants/$v/libkrepp_amd.so krepp_amd/lib/libkrepp_amd.so; one $v X=1 $B; done
cp /tmp/main_lib.so krepp_amd/lib/libkrepp_amd.so
one head_b X=1 $B

#### session r ############################################################
#!/bin/bash
# round 3, session R: the whole GPU suite and the profile passes on the final code; traffic_latest.json
ulimit -c 0
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --durations=6 > gpurun_out/r3r_pytest_gpu.log 2>&1; echo "rc=$?" >> gpurun_out/r3r_pytest_gpu.log
tail -10 gpurun_out/r3r_pytest_gpu.log
rm -rf /tmp/pytest-of-* /tmp/krepp_*
python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -2
bash scripts/profile.sh r3b > gpurun_out/prof_r3b.log 2>&1
tail -3 gpurun_out/prof_r3b.log
rm -rf /tmp/krepp_bench_*
python bench.py --steps 10 --warmup 2 > gpurun_out/r3r_bench.json 2> gpurun_out/r3r_bench.err
python scripts/traffic.py gpurun_out/prof_r3b gpurun_out/r3r_bench.json gpurun_out/traffic_r3b.json
tail -c 600 gpurun_out/r3r_bench.json

#### session s ############################################################
#!/bin/bash
# Round 3, session S: do the scan's launch-time levels follow the GPU's clock levels?  (a) what the driver publishes, sampled
# next to a bench run whose streams are run again after idle gaps; (b) rocm-smi's view before and after.
ulimit -c 0
cd "$(dirname "$0")/.."
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out
ls -la /sys/class/drm/ 2>&1 | head -20
ls /sys/class/drm/card*/device/ 2>&1 | head -80
rocm-smi --showclocks --showpower --showperflevel --showtemp 2>&1 | head -60
python3 scripts/clock_probe.py 0.02 python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-host-inclusive \
   --stream-variance 5 --stream-variance-idle 1.0 > gpurun_out/r3s_clocks.txt 2> gpurun_out/r3s_bench.err
grep "stream-variance" gpurun_out/r3s_bench.err
head -5 gpurun_out/r3s_clocks.txt; wc -l gpurun_out/r3s_clocks.txt
# rocm-smi next to a second run (in case sysfs is not readable): one sample every ~0.3 s
python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-host-inclusive --stream-variance 4 --stream-variance-idle 1.0 \
   > gpurun_out/r3s_bench2.json 2> gpurun_out/r3s_bench2.err &
BP=$!
while kill -0 $BP 2>/dev/null; do
  echo "t=$(date +%s.%N)"; rocm-smi --showclocks --showpower 2>&1 | grep -E "clk|Power|power" ; sleep 0.1
done > gpurun_out/r3s_smi.txt
wait $BP
grep "stream-variance" gpurun_out/r3s_bench2.err
tail -40 gpurun_out/r3s_smi.txt
python3 -m pytest tests/test_gpu_long_sequences.py -x -q 2>&1 | tail -3
# sequences between a read and a contig: where should tiling start?
for L in 400 600 1000; do
  NC=$((30000000 / L))
  echo "== $L bp x $NC, tiles from 1,024 positions (default: none of these are tiled)"; python3 scripts/time_contigs.py $L $NC 2>&1 | tail -3
  echo "== $L bp x $NC, tiles from 256 positions"; KR_TILE_MIN_POS=256 python3 scripts/time_contigs.py $L $NC 2>&1 | tail -3
done

#### session t ############################################################
#!/bin/bash
# Round 3, session T: which of a stream's buffers decides the scan's launch-time level?  (kr_debug_stream_move) + syn10000 on the final code
ulimit -c 0
cd "$(dirname "$0")/.."
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out
for rep in 1 2; do
  python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-host-inclusive --stream-variance 5 --stream-variance-move 4,2,1,0,3 \
     > gpurun_out/r3t_move$rep.json 2> gpurun_out/r3t_move$rep.err
  grep "stream-variance" gpurun_out/r3t_move$rep.err
done
rm -rf /tmp/krepp_bench_*
python3 bench.py --workload syn10000 --steps 10 --warmup 2 --no-cpu-baseline --no-host-inclusive > gpurun_out/r3t_syn10000.json 2> gpurun_out/r3t_syn10000.err
python3 - <<'PY'
import json
d = json.loads(open('gpurun_out/r3t_syn10000.json').read().strip().splitlines()[-1])
print('syn10000:', round(d['value'] / 1e6, 2), round(d['ms_per_step'], 2), {k: round(v, 2) for k, v in d['kernel_ms'].items() if isinstance(v, float)}, d['check']['rows_equal'])
PY

#### session u ############################################################
#!/bin/bash
# Round 3, session U: addresses of a stream's buffers next to its scan level; does every move of the item list / the counters draw a new level?
ulimit -c 0
cd "$(dirname "$0")/.."
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out
for rep in 1 2 3; do
  python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-host-inclusive --stream-variance 6 --stream-variance-move 2,0,2,0 \
     > gpurun_out/r3u_move$rep.json 2> gpurun_out/r3u_move$rep.err
  grep "stream-variance" gpurun_out/r3u_move$rep.err
  tail -2 gpurun_out/r3u_move$rep.err | cut -c1-200
done

#### session v ############################################################
#!/bin/bash
# Round 3, session V: what the waves of a slow scan launch wait for -- counter passes over ONE process each with fast and slow streams in it
ulimit -c 0
cd "$(dirname "$0")/.."
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out
pass() {
  name=$1; shift
  OUT=$PWD/gpurun_out/r3v_$name
  rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-inclusive --read-procs 1 --check-reads 2000 --distinct-batches 1 --stream-variance 7 > $OUT/bench.log 2>&1
  echo "== $name: $*"
  python3 - $OUT <<'PY'
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)
if not f:
    print('no counter file'); print(open(sys.argv[1] + '/bench.log').read()[-1500:]); sys.exit(0)
rows = collections.OrderedDict()
for r in csv.DictReader(open(f[0])):
    if 'kr_scan' not in r['Kernel_Name']: continue
    k = int(r['Dispatch_Id'])
    d = rows.setdefault(k, {'dur': (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6})
    d[r['Counter_Name']] = d.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
for k, v in rows.items():
    if v['dur'] < 5: continue
    print(k, {a: (round(b, 2) if a == 'dur' else f'{b:.5g}') for a, b in v.items()})
PY
  rm -rf /tmp/krepp_bench_* 
}
pass icache SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL
pass wait SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES
pass insts SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_LDS
pass wr TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum
pass atomic TCC_EA0_ATOMIC_sum TCC_EA0_ATOMIC_LEVEL_sum TCC_ATOMIC_sum SQC_DCACHE_MISSES SQC_DCACHE_REQ
pass cyc SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS

#### session w ############################################################
#!/bin/bash
# Round 3, session W: do physically contiguous allocations (hipDeviceMallocContiguous) remove the scan's slow launch levels?
ulimit -c 0
cd "$(dirname "$0")/.."
export GPU_MAX_HW_QUEUES=8 KR_HBM_VERBOSE=1
mkdir -p gpurun_out
for mode in 3 0 3 2 1 3 0; do
  KR_HBM_CONTIGUOUS=$mode python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-host-inclusive --stream-variance 6 \
     > gpurun_out/r3w_m$mode.json 2> gpurun_out/r3w_m$mode.err
  echo "== KR_HBM_CONTIGUOUS=$mode"
  python3 - gpurun_out/r3w_m$mode.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('timed steps:', round(d['value'] / 1e6, 2), 'M reads/s', {k: round(v, 2) for k, v in d['kernel_ms'].items() if isinstance(v, float)}, d['check']['rows_equal'])
PY
  grep -E "stream-variance\] stream [0-9]+:|no contiguous" gpurun_out/r3w_m$mode.err | sed 's/\[stream-variance\] //'
  rm -rf /tmp/krepp_bench_*
done

#### session x ############################################################
#!/bin/bash
# Round 3, session X: address-translation counters of the scan with default and with physically contiguous stream buffers
ulimit -c 0
cd "$(dirname "$0")/.."
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out
pass() {
  name=$1; mode=$2; shift; shift
  OUT=$PWD/gpurun_out/r3x_$name
  rm -rf $OUT; mkdir -p $OUT
  export KR_HBM_CONTIGUOUS=$mode
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-inclusive --read-procs 1 --check-reads 2000 --distinct-batches 1 --stream-variance 6 > $OUT/bench.log 2>&1
  echo "== $name (KR_HBM_CONTIGUOUS=$mode): $*"
  python3 - $OUT <<'PY'
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)
if not f:
    print('no counter file'); print(open(sys.argv[1] + '/bench.log').read()[-1500:]); sys.exit(0)
rows = collections.OrderedDict()
for r in csv.DictReader(open(f[0])):
    kn = r['Kernel_Name']
    tag = 'scan' if 'kr_scan' in kn else ('acc' if 'kr_acc_kernel_t<true, 5, false, 7>' in kn else ('select' if 'kr_select' in kn else None))
    if not tag: continue
    k = (int(r['Dispatch_Id']), tag)
    d = rows.setdefault(k, {'dur': (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6})
    d[r['Counter_Name']] = d.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
for k, v in rows.items():
    if v['dur'] < 3: continue
    print(k[0], k[1], ' '.join(f"{a.replace('TCP_UTCL1_', '').replace('_sum', '')}={(round(b, 2) if a == 'dur' else format(b, '.4g'))}" for a, b in v.items()))
PY
  rm -rf /tmp/krepp_bench_*
}
C1="TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_THRASHING_STALL_sum"
C2="TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_SERIALIZATION_STALL_sum"
pass tlb_default 0 $C1
pass tlb_contig 2 $C1
pass stall_default 0 $C2
pass stall_contig 2 $C2

#### session y ############################################################
#!/bin/bash
# Round 3, session Y: select kernel three loads deep across reads, dedup kernel with one 16-byte probe per slot; larger batches
ulimit -c 0
cd "$(dirname "$0")/.."
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out
python3 -m pytest tests/test_gpu_parity.py -x -q -k "golden or report_modes or large_batch or device_brent or where_a_streams or lanes" 2>&1 | tail -3
python3 -m pytest tests/test_gpu_syn1000.py -x -q -k "slotted" 2>&1 | tail -3
rm -rf /tmp/pytest-of-* /tmp/krepp_*
trace() {
  OUT=$PWD/gpurun_out/r3y_$1; rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-inclusive --read-procs 1 --distinct-batches 1 > $OUT/bench.log 2>&1
  echo "== $1"; python3 scripts/kstats.py $OUT | grep -E "select|dedup_kernel|llh_kernel|acc_kernel_t<true, 5, false, 7>|scan_pipe" | cut -c1-200
  rm -rf /tmp/krepp_bench_*
}
trace new
cp krepp_amd/lib/libkrepp_amd.so /tmp/main_lib.so
cp krepp_amd/lib/variants/probe8/libkrepp_amd.so krepp_amd/lib/libkrepp_amd.so
trace probe8
cp /tmp/main_lib.so krepp_amd/lib/libkrepp_amd.so
for n in 8000000 12000000 16000000; do
  python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-inclusive --reads-per-step $n > gpurun_out/r3y_n$n.json 2> gpurun_out/r3y_n$n.err
  python3 - gpurun_out/r3y_n$n.json $n <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
n = int(sys.argv[2]) / 1e6
print(sys.argv[2], 'reads per launch:', round(d['value'] / 1e6, 2), 'M reads/s', {k: round(v / n, 3) for k, v in d['kernel_ms'].items() if isinstance(v, float) and k in ('scan', 'accumulate', 'llh_select')}, 'ms per million reads', d['check']['rows_equal'])
PY
  rm -rf /tmp/krepp_bench_*
done

#### session z ############################################################
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

#### session aa ############################################################
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

#### session ab ############################################################
#!/bin/bash
# Round 3, session AB: whole parity file on the current code (select as at the start of the day, dedup with the 16-byte probe), kernel times
ulimit -c 0
cd "$(dirname "$0")/.."
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_long_sequences.py tests/test_place.py -x -q -m gpu 2>&1 | tail -5 | cut -c1-300
rm -rf /tmp/pytest-of-* /tmp/krepp_*
OUT=$PWD/gpurun_out/r3ab_trace; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-inclusive --read-procs 1 --distinct-batches 1 > $OUT/bench.log 2>&1
python3 scripts/kstats.py $OUT | grep -E "select|dedup_kernel|llh_kernel|scan_pipe|acc_kernel_t<true, 5, false, 7>" | cut -c1-200
rm -rf /tmp/krepp_bench_*
python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-host-inclusive 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('reads/s', round(d['value']/1e6,2), 'ms/step', round(d['ms_per_step'],2), {k: round(v,2) for k,v in d['kernel_ms'].items() if isinstance(v,float)}, d['check'])"

#### session ac ############################################################
#!/bin/bash
ulimit -c 0
cd "$(dirname "$0")/.."
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out
echo "== where_a_streams + lanes"
python3 -m pytest tests/test_gpu_parity.py -x -q -vv -k "where_a_streams or lanes" 2>&1 | grep -v "^$" | tail -60 | cut -c1-400
echo "== everything before the new test + lanes (new test deselected)"
python3 -m pytest tests/test_gpu_parity.py -x -q -k "not where_a_streams" 2>&1 | tail -5 | cut -c1-300

#### session ad ############################################################
#!/bin/bash
ulimit -c 0
cd "$(dirname "$0")/.."
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8
for part in c 0 1 2 3 4 x; do
  echo "== part $part"
  KR_TEST_PART=$part python3 -m pytest tests/test_gpu_parity.py -x -q -k "where_a_streams or lanes" 2>&1 | grep -E "passed|failed|At index" | cut -c1-200
done

#### session ae ############################################################
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

#### session af ############################################################
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

#### session ag ############################################################
#!/bin/bash
# round 3, session AG: the whole GPU suite and the profile passes on the final code; traffic_latest.json
ulimit -c 0
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --durations=6 > gpurun_out/r3ag_pytest_gpu.log 2>&1; echo "rc=$?" >> gpurun_out/r3ag_pytest_gpu.log
tail -10 gpurun_out/r3ag_pytest_gpu.log
rm -rf /tmp/pytest-of-* /tmp/krepp_*
python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -2
bash scripts/profile.sh r3c > gpurun_out/prof_r3c.log 2>&1
tail -3 gpurun_out/prof_r3c.log
rm -rf /tmp/krepp_bench_*
python bench.py --steps 10 --warmup 2 > gpurun_out/r3ag_bench.json 2> gpurun_out/r3ag_bench.err
python scripts/traffic.py gpurun_out/prof_r3c gpurun_out/r3ag_bench.json gpurun_out/traffic_r3c.json
tail -c 600 gpurun_out/r3ag_bench.json

#### session ah ############################################################
#!/bin/bash
# Round 3, session AH: a stream tries a few allocations of its item list and keeps the one the scan ran fastest on
ulimit -c 0
cd "$(dirname "$0")/.."
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out
python3 -m pytest tests/test_gpu_syn1000.py tests/test_gpu_parity.py -x -q -k "slotted or large_batch or golden or lanes or where_a" 2>&1 | tail -3 | cut -c1-200
rm -rf /tmp/pytest-of-* /tmp/krepp_*
run() {
  echo "== $1 (KR_ITEM_PLACEMENT_TRIALS=$2)"
  KR_ITEM_PLACEMENT_TRIALS=$2 KR_ITEM_PLACEMENT_VERBOSE=1 python3 bench.py --steps 6 --no-cpu-baseline --no-host-inclusive > gpurun_out/r3ah_$1.json 2> gpurun_out/r3ah_$1.err
  grep "item list" gpurun_out/r3ah_$1.err | sed 's/\[krepp_amd\] //' | tr '\n' ';'; echo
  python3 - gpurun_out/r3ah_$1.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('  ', round(d['value'] / 1e6, 2), 'M reads/s', round(d['ms_per_step'], 2), 'ms/step', {k: round(v, 2) for k, v in d['kernel_ms'].items() if isinstance(v, float)}, d['config']['item_list_placement'], d['check']['rows_equal'])
PY
  rm -rf /tmp/krepp_bench_*
}
for i in 1 2 3 4 5 6; do run on$i 3; done
for i in 1 2 3; do run off$i 0; done

#### session ai ############################################################
#!/bin/bash
# Round 3, session AI: placement trials under the tests that use large batches; the default bench (host-inclusive leg included); two ranks on one GPU
ulimit -c 0
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
python3 -m pytest tests/test_gpu_syn1000.py tests/test_gpu_bench.py -x -q --durations=4 2>&1 | tail -9 | cut -c1-200
rm -rf /tmp/pytest-of-* /tmp/krepp_*
KR_ITEM_PLACEMENT_VERBOSE=1 python3 bench.py --steps 10 > gpurun_out/r3ai_bench.json 2> gpurun_out/r3ai_bench.err
grep -c "item list" gpurun_out/r3ai_bench.err
python3 - <<'PY'
import json
d = json.loads(open('gpurun_out/r3ai_bench.json').read().strip().splitlines()[-1])
print(round(d['value'] / 1e6, 2), 'M reads/s', round(d['ms_per_step'], 2), 'ms/step', {k: round(v, 2) for k, v in d['kernel_ms'].items() if isinstance(v, float)}, d['config']['item_list_placement'], d['check']['rows_equal'])
print('host-inclusive', round(d['value_host_inclusive']['value'] / 1e6, 2), 'M reads/s; warmup', d['warmup'], 'frac', round(d['roofline']['frac'], 3), 'traffic_frac', d['roofline'].get('traffic_frac'))
PY

#### session aj ############################################################
#!/bin/bash
# Round 3, session AJ: per-launch scan times of the timed steps after the placement trials (does the kept list keep its level?)
ulimit -c 0
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
for i in 1 2 3; do
  KR_ITEM_PLACEMENT_VERBOSE=1 python3 bench.py --steps 10 --no-cpu-baseline --no-host-inclusive > gpurun_out/r3aj_$i.json 2> gpurun_out/r3aj_$i.err
  grep "item list" gpurun_out/r3aj_$i.err | sed 's/\[krepp_amd\] item list //' | tr '\n' ';'; echo
  python3 - gpurun_out/r3aj_$i.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('  ', round(d['value'] / 1e6, 2), 'M reads/s', d['kernel_ms']['scan_per_launch'], d['config']['item_list_placement'])
PY
  rm -rf /tmp/krepp_bench_*
done

#### session ak ############################################################
#!/bin/bash
# Round 3, session AK: scan launch times around a large allocation and a large free (no placement trials)
ulimit -c 0
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
KR_ITEM_PLACEMENT_TRIALS=0 python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-inclusive --churn-gb 48 > gpurun_out/r3ak.json 2> gpurun_out/r3ak.err
grep churn gpurun_out/r3ak.err
python3 - <<'PY'
import json
d = json.loads(open('gpurun_out/r3ak.json').read().strip().splitlines()[-1])
print(d['kernel_ms']['scan_per_launch'])
PY

#### session al ############################################################
#!/bin/bash
# round 3, session AL: the whole GPU suite and the profile passes on the final code; traffic_latest.json
ulimit -c 0
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --durations=6 > gpurun_out/r3al_pytest_gpu.log 2>&1; echo "rc=$?" >> gpurun_out/r3al_pytest_gpu.log
tail -10 gpurun_out/r3al_pytest_gpu.log
rm -rf /tmp/pytest-of-* /tmp/krepp_*
python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -2
bash scripts/profile.sh r3d > gpurun_out/prof_r3d.log 2>&1
tail -3 gpurun_out/prof_r3d.log
rm -rf /tmp/krepp_bench_*
python bench.py --steps 10 --warmup 2 > gpurun_out/r3al_bench.json 2> gpurun_out/r3al_bench.err
python scripts/traffic.py gpurun_out/prof_r3d gpurun_out/r3al_bench.json gpurun_out/traffic_r3d.json
tail -c 600 gpurun_out/r3al_bench.json

#### session am ############################################################
#!/bin/bash
# Round 3, session AM: what the management interface says (throttle status, clocks, temperatures) while the scan switches levels
ulimit -c 0
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
which amd-smi rocm-smi
amd-smi metric --help 2>&1 | head -30
(amd-smi metric -g 0 --json 2>&1 | head -150) > gpurun_out/r3am_idle_metric.txt
KR_ITEM_PLACEMENT_TRIALS=0 python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-host-inclusive --churn-gb 1 > gpurun_out/r3am.json 2> gpurun_out/r3am.err &
BP=$!
sleep 20
while kill -0 $BP 2>/dev/null; do
  echo "t=$(date +%s.%N)"
  rocm-smi --showmetrics 2>/dev/null | grep -i -E "throttle|uclk|fclk|gfxclk|socket_power|temperature_hotspot|temperature_mem|hbm|current_socclk|indep_throttle|prochot|ppt|thm" | head -40
  sleep 0.2
done > gpurun_out/r3am_smi.txt
wait $BP
grep churn gpurun_out/r3am.err | cut -c1-600
wc -l gpurun_out/r3am_smi.txt; head -60 gpurun_out/r3am_smi.txt

#### session ao ############################################################
#!/bin/bash
# round 3, session AO: last check of the round -- the whole GPU suite, smoke, the default bench as the driver runs it
ulimit -c 0
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --durations=5 > gpurun_out/r3ao_pytest_gpu.log 2>&1; echo "rc=$?" >> gpurun_out/r3ao_pytest_gpu.log
tail -9 gpurun_out/r3ao_pytest_gpu.log
rm -rf /tmp/pytest-of-* /tmp/krepp_*
python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -2
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3ao_bench.json 2> gpurun_out/r3ao_bench.err
python3 - <<'PY'
import json
d = json.loads(open('gpurun_out/r3ao_bench.json').read().strip().splitlines()[-1])
print(round(d['value'] / 1e6, 2), 'M reads/s', round(d['ms_per_step'], 2), 'ms/step', {k: (round(v, 2) if isinstance(v, float) else v) for k, v in d['kernel_ms'].items() if k != 'source'})
print(d['config']['item_list_placement'], 'host-inclusive', round(d['value_host_inclusive']['value'] / 1e6, 2), 'frac', round(d['roofline']['frac'], 3), 'traffic', d['roofline']['traffic'], d['roofline']['traffic_source'].get('commit'), d['check'])
PY

#### session ap ############################################################
#!/bin/bash
# round 3, session AP: the scan's episodes (40.9 <-> 44.4 ms within one stream) with and without the other kernels around it
ulimit -c 0
mkdir -p gpurun_out
export KR_ITEM_PLACEMENT_TRIALS=0
run() {
  name=$1; shift
  env "$@" > gpurun_out/r3ap_$name.json 2> gpurun_out/r3ap_$name.err
  python3 - gpurun_out/r3ap_$name.json $name <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], round(d['ms_per_step'], 2), 'ms/step; scan per launch:', d['kernel_ms']['scan_per_launch'])
PY
  rm -rf /tmp/krepp_bench_*
}
B="python3 bench.py --steps 40 --warmup 2 --no-cpu-baseline --no-host-inclusive --check-reads 2000"
run default X=1 $B
run scan_only KR_DEBUG_SKIP=2 $B
run two_streams X=1 $B --pipeline-streams 2
run default_again X=1 $B

#### session aq ############################################################
#!/bin/bash
# round 3, session AQ: accumulate kernel with the next read's metadata waiting in LDS
ulimit -c 0
mkdir -p gpurun_out
export KR_ITEM_PLACEMENT_TRIALS=0
python3 -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3 | cut -c1-200
rm -rf /tmp/pytest-of-* /tmp/krepp_*
run() {
  python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-host-inclusive > gpurun_out/r3aq_$1.json 2> gpurun_out/r3aq_$1.err
  python3 - gpurun_out/r3aq_$1.json $1 <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], round(d['value'] / 1e6, 2), 'M reads/s', {k: round(v, 2) for k, v in d['kernel_ms'].items() if isinstance(v, float)}, d['check']['rows_equal'])
PY
  rm -rf /tmp/krepp_bench_*
}
cp krepp_amd/lib/libkrepp_amd.so /tmp/main_lib.so
run ahead_a
cp krepp_amd/lib/variants/meta0/libkrepp_amd.so krepp_amd/lib/libkrepp_amd.so
run off_a
cp /tmp/main_lib.so krepp_amd/lib/libkrepp_amd.so
run ahead_b
cp krepp_amd/lib/variants/meta0/libkrepp_amd.so krepp_amd/lib/libkrepp_amd.so
run off_b
cp /tmp/main_lib.so krepp_amd/lib/libkrepp_amd.so

#### session ar ############################################################
#!/bin/bash
# round 3, session AR: select kernel with two reads per pass of a group of lanes
ulimit -c 0
export TMPDIR=/tmp KR_ITEM_PLACEMENT_TRIALS=0
mkdir -p gpurun_out
python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_long_sequences.py -x -q 2>&1 | tail -3 | cut -c1-200
python3 -m pytest tests/test_gpu_syn1000.py -x -q -k "slotted" 2>&1 | tail -2 | cut -c1-200
rm -rf /tmp/pytest-of-* /tmp/krepp_*
trace() {
  OUT=$PWD/gpurun_out/r3ar_$1; rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-inclusive --read-procs 1 --distinct-batches 1 > $OUT/bench.log 2>&1
  echo "== $1"; python3 scripts/kstats.py $OUT | grep -E "select|dedup_kernel" | cut -c1-200
  grep -h '"metric"' $OUT/bench.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('   rows_equal', d['check']['rows_equal'], 'llh_select', round(d['kernel_ms']['llh_select'],2))"
  rm -rf /tmp/krepp_bench_*
}
trace pair
cp krepp_amd/lib/libkrepp_amd.so /tmp/main_lib.so
cp krepp_amd/lib/variants/pair0/libkrepp_amd.so krepp_amd/lib/libkrepp_amd.so
trace single
cp /tmp/main_lib.so krepp_amd/lib/libkrepp_amd.so

#### session as ############################################################
#!/bin/bash
# round 3, session AS: syn10000 accumulate time with and without the two-segment launch in the chain
ulimit -c 0
mkdir -p gpurun_out
export KR_ITEM_PLACEMENT_TRIALS=0
run() {
  name=$1; shift
  env "$@" python3 bench.py --workload syn10000 --steps 8 --warmup 2 --no-cpu-baseline --no-host-inclusive > gpurun_out/r3as_$name.json 2> gpurun_out/r3as_$name.err
  python3 - gpurun_out/r3as_$name.json $name <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], round(d['value'] / 1e6, 2), 'M reads/s', {k: round(v, 2) for k, v in d['kernel_ms'].items() if isinstance(v, float)}, d['check']['rows_equal'])
PY
  rm -rf /tmp/krepp_bench_*
}
run default X=1
run no_lean2 KR_DEBUG_NO_LEAN2=1
run default_b X=1
run no_lean2_b KR_DEBUG_NO_LEAN2=1

#### session at ############################################################
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

