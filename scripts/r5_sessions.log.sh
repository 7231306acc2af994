#!/bin/bash
# The GPU sessions of round 5 as they were run (gpurun -- 'bash scripts/r5_sN_session.sh'), folded into one file: a log, not a tool.
# Each block was a script of its own; outputs went to gpurun_out/ and, where they are quoted, to profiles/round5_*.

##################### session 1 #####################
# round 5, session 1: the three verification tests of the round-4 review + this box's baseline lines + the accumulate kernel's
# per-read statistics on both synthetic indexes (the KR_STATS build: scripts/build_variant.sh stats -DKR_STATS=1)
ulimit -c 0
mkdir -p gpurun_out
python -m pytest tests/test_gpu_syn1000.py tests/test_gpu_bench.py::test_bench_eight_ranks_share_one_gpu -x -q -s --durations=8 > gpurun_out/r5_s1_tests.txt 2>&1
tail -25 gpurun_out/r5_s1_tests.txt
B="--no-cpu-baseline --no-host-inclusive --steps 8 --warmup 2 --check-reads 2000 --skip-host-path-check"
python bench.py $B > gpurun_out/r5_s1_syn1000.json 2> gpurun_out/r5_s1_syn1000.err
python bench.py --workload syn10000 $B > gpurun_out/r5_s1_syn10000.json 2> gpurun_out/r5_s1_syn10000.err
cp krepp_amd/lib/libkrepp_amd.so /tmp/main_lib.so
cp krepp_amd/lib/variants/stats/libkrepp_amd.so krepp_amd/lib/libkrepp_amd.so
S="--no-cpu-baseline --no-host-inclusive --steps 1 --warmup 0 --check-reads 1000 --skip-host-path-check --distinct-batches 1"
KR_ITEM_PLACEMENT_TRIALS=0 KR_DEBUG_SKIP=512 python bench.py --workload syn10000 $S > gpurun_out/r5_s1_stats10000.json 2> gpurun_out/r5_s1_stats10000.err
KR_ITEM_PLACEMENT_TRIALS=0 KR_DEBUG_SKIP=512 python bench.py $S > gpurun_out/r5_s1_stats1000.json 2> gpurun_out/r5_s1_stats1000.err
cp /tmp/main_lib.so krepp_amd/lib/libkrepp_amd.so
for f in syn1000 syn10000; do python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/r5_s1_$f.json') if l.startswith('{')][-1]); print('$f', round(d['value']/1e6,2), {k:(round(x,2) if isinstance(x,float) else x) for k,x in d['kernel_ms'].items() if k in ('scan','accumulate','llh_select')}, d['check']['rows_equal'])"; done
grep -h "kr stats" gpurun_out/r5_s1_stats10000.err | tail -8
grep -h "kr stats" gpurun_out/r5_s1_stats1000.err | tail -8

##################### session 2 #####################
# round 5, session 2: the device-side report text (kr_dev_text.inc): parity tests, then the CLI end to end on both indexes
ulimit -c 0
mkdir -p gpurun_out
python -m pytest tests/test_gpu_text.py tests/test_place.py::test_cli_dist_on_a_file_large_enough_for_the_parallel_reader tests/test_gpu_rccl_cli.py -x -q --durations=5 > gpurun_out/r5_s2_tests.txt 2>&1
tail -15 gpurun_out/r5_s2_tests.txt
python scripts/time_cli.py 16000000 > gpurun_out/r5_s2_cli_toy25.txt 2>&1
grep -v "^place" gpurun_out/r5_s2_cli_toy25.txt | head -40
python scripts/time_cli_syn1000.py 8e6 > gpurun_out/r5_s2_cli_syn1000.txt 2>&1
cat gpurun_out/r5_s2_cli_syn1000.txt

##################### session 3 #####################
# round 5, session 3: direct-mapped likelihood de-duplication (parity + A/B on both indexes), CLI with the initialisation / batch
# loop / tear-down split and parallel pwrite
ulimit -c 0
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_syn1000.py::test_syn1000_10gb_index_vs_oracle_and_full_batch_properties --deselect tests/test_gpu_bench.py --durations=5 > gpurun_out/r5_s3_tests.txt 2>&1
tail -12 gpurun_out/r5_s3_tests.txt
B="--no-cpu-baseline --no-host-inclusive --steps 8 --warmup 2 --check-reads 2000 --skip-host-path-check"
for w in syn1000 syn10000; do for d in 1 0; do
  KR_DD_DIRECT=$d python bench.py --workload $w $B > gpurun_out/r5_s3_${w}_dd$d.json 2> gpurun_out/r5_s3_${w}_dd$d.err
  python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/r5_s3_${w}_dd$d.json') if l.startswith('{')][-1]); print('$w direct=$d', round(d['value']/1e6,2), {k:(round(x,2) if isinstance(x,float) else x) for k,x in d['kernel_ms'].items() if k in ('scan','accumulate','llh_select')}, d['check']['rows_equal'])"
done; done
python scripts/time_cli.py 16000000 > gpurun_out/r5_s3_cli_toy25.txt 2>&1
grep -v "^place" gpurun_out/r5_s3_cli_toy25.txt | head -30
KR_TIME_CLI_CONFIGS=0,2,3,4,6,8 python scripts/time_cli_syn1000.py 8e6 > gpurun_out/r5_s3_cli_syn1000.txt 2>&1
cat gpurun_out/r5_s3_cli_syn1000.txt

##################### session 4 #####################
# round 5, session 4: CLI with detached batches / multi-chunk batches / exit without tear-down; per-kernel times with the direct
# de-duplication on both indexes
ulimit -c 0
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_syn1000.py::test_syn1000_10gb_index_vs_oracle_and_full_batch_properties --deselect tests/test_gpu_bench.py --durations=5 > gpurun_out/r5_s4_tests.txt 2>&1
tail -9 gpurun_out/r5_s4_tests.txt
python scripts/time_cli.py 16000000 > gpurun_out/r5_s4_cli_toy25.txt 2>&1
grep "^dist" gpurun_out/r5_s4_cli_toy25.txt | head -30
KR_TIME_CLI_CONFIGS=0,3,4,6,8 python scripts/time_cli_syn1000.py 8e6 > gpurun_out/r5_s4_cli_syn1000.txt 2>&1
cat gpurun_out/r5_s4_cli_syn1000.txt
bash scripts/ktimes.sh s10k --workload syn10000 > gpurun_out/r5_s4_ktimes_syn10000.txt 2>&1
grep -v "relayout\|build_" gpurun_out/r5_s4_ktimes_syn10000.txt
bash scripts/ktimes.sh s1k > gpurun_out/r5_s4_ktimes_syn1000.txt 2>&1
grep -v "relayout\|build_" gpurun_out/r5_s4_ktimes_syn1000.txt
KR_DD_DIRECT=0 bash scripts/ktimes.sh s10k0 --workload syn10000 > gpurun_out/r5_s4_ktimes_syn10000_dd0.txt 2>&1
grep "dedup\|select\|llh" gpurun_out/r5_s4_ktimes_syn10000_dd0.txt

##################### session 5 #####################
# round 5, session 5: CLI after the reader fixes; list-position chunk size and direct de-duplication A/B by kernel time; a full
# default bench line (host-inclusive leg) on this box
ulimit -c 0
mkdir -p gpurun_out
python -m pytest tests/test_place.py tests/test_gpu_rccl_cli.py tests/test_gpu_text.py tests/test_seek.py -m gpu -x -q > gpurun_out/r5_s5_tests.txt 2>&1
tail -4 gpurun_out/r5_s5_tests.txt
python scripts/time_cli.py 16000000 > gpurun_out/r5_s5_cli_toy25.txt 2>&1
grep "^dist" gpurun_out/r5_s5_cli_toy25.txt | head -30
KR_TIME_CLI_CONFIGS=0,4,6,7,8 python scripts/time_cli_syn1000.py 8e6 > gpurun_out/r5_s5_cli_syn1000.txt 2>&1
cat gpurun_out/r5_s5_cli_syn1000.txt
cp krepp_amd/lib/libkrepp_amd.so /tmp/main_lib.so
for v in rc16 rc128; do
  cp krepp_amd/lib/variants/$v/libkrepp_amd.so krepp_amd/lib/libkrepp_amd.so
  bash scripts/ktimes.sh $v --workload syn10000 > gpurun_out/r5_s5_ktimes_$v.txt 2>&1
  echo "== $v"; grep "dedup\|select\|llh\|acc_kernel_t<true, 5, false, 7\|scan_pipe" gpurun_out/r5_s5_ktimes_$v.txt
done
cp /tmp/main_lib.so krepp_amd/lib/libkrepp_amd.so
KR_DD_DIRECT=0 bash scripts/ktimes.sh dd0 --workload syn10000 > gpurun_out/r5_s5_ktimes_dd0.txt 2>&1
echo "== direct off"; grep "dedup\|select\|llh" gpurun_out/r5_s5_ktimes_dd0.txt
bash scripts/ktimes.sh s1k > gpurun_out/r5_s5_ktimes_syn1000.txt 2>&1
echo "== syn1000"; grep -v "relayout\|build_" gpurun_out/r5_s5_ktimes_syn1000.txt
KR_DD_DIRECT=0 bash scripts/ktimes.sh s1k0 > gpurun_out/r5_s5_ktimes_syn1000_dd0.txt 2>&1
echo "== syn1000 direct off"; grep "dedup\|select\|llh" gpurun_out/r5_s5_ktimes_syn1000_dd0.txt
KR_ITEM_PLACEMENT_TRIALS=0 KR_DEBUG_SKIP=512 python bench.py --workload syn10000 --no-cpu-baseline --no-host-inclusive --steps 1 --warmup 0 --check-reads 1000 --skip-host-path-check --distinct-batches 1 2>&1 | grep "kr stats" | head -2
python bench.py --no-cpu-baseline --steps 10 > gpurun_out/r5_s5_bench_default.json 2> gpurun_out/r5_s5_bench_default.err
python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/r5_s5_bench_default.json') if l.startswith('{')][-1]); print(round(d['value']/1e6,2), d['kernel_ms']['scan_per_launch'], json.dumps(d['value_host_inclusive'])[:1500], d['config']['item_list_placement'])"

##################### session 6 #####################
# round 5, session 6: list-position chunks (256 adaptive, 1024), direct de-duplication on/off by kernel time; the 10,000-genome
# index at 4 M and 8 M reads per step; which direction of PCIe traffic slows the scan in the host-inclusive leg
ulimit -c 0
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_text.py tests/test_gpu_syn1000.py::test_ten_thousand_genome_index_vs_oracle tests/test_gpu_long_sequences.py -x -q > gpurun_out/r5_s6_tests.txt 2>&1
tail -4 gpurun_out/r5_s6_tests.txt
for w in syn10000 syn1000; do
  bash scripts/ktimes.sh ${w}_main --workload $w > gpurun_out/r5_s6_ktimes_${w}_main.txt 2>&1
  echo "== $w main"; grep "dedup\|select\|llh" gpurun_out/r5_s6_ktimes_${w}_main.txt
  KR_DD_DIRECT=0 bash scripts/ktimes.sh ${w}_dd0 --workload $w > gpurun_out/r5_s6_ktimes_${w}_dd0.txt 2>&1
  echo "== $w direct off"; grep "dedup\|select\|llh" gpurun_out/r5_s6_ktimes_${w}_dd0.txt
done
cp krepp_amd/lib/libkrepp_amd.so /tmp/main_lib.so
cp krepp_amd/lib/variants/rc1024/libkrepp_amd.so krepp_amd/lib/libkrepp_amd.so
bash scripts/ktimes.sh rc1024 --workload syn10000 > gpurun_out/r5_s6_ktimes_rc1024.txt 2>&1
echo "== rc1024"; grep "dedup\|select\|llh" gpurun_out/r5_s6_ktimes_rc1024.txt
cp /tmp/main_lib.so krepp_amd/lib/libkrepp_amd.so
B="--no-cpu-baseline --no-host-inclusive --steps 6 --warmup 2 --check-reads 2000 --skip-host-path-check"
for n in 2000000 4000000 8000000; do
  python bench.py --workload syn10000 --reads-per-step $n $B > gpurun_out/r5_s6_syn10000_$n.json 2> gpurun_out/r5_s6_syn10000_$n.err
  python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/r5_s6_syn10000_$n.json') if l.startswith('{')][-1]); print('syn10000 reads/step $n', round(d['value']/1e6,2), {k:(round(x,2) if isinstance(x,float) else x) for k,x in d['kernel_ms'].items() if k in ('scan','accumulate','llh_select')}, d['check']['rows_equal'])"
done
for v in no_d2h no_h2d; do
  python bench.py --no-cpu-baseline --steps 4 --warmup 2 --check-reads 2000 --skip-host-path-check --host-leg-variant $v > gpurun_out/r5_s6_hostleg_$v.json 2> gpurun_out/r5_s6_hostleg_$v.err
  python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/r5_s6_hostleg_$v.json') if l.startswith('{')][-1]); h=d['value_host_inclusive']; print('$v', round(d['value']/1e6,2), round(h['value']/1e6,2), round(h['steady_state']['value']/1e6,2), h['kernel_ms_in_this_leg'])"
done

##################### session 7 #####################
# round 5, session 7: record slots in larger chunks (the accumulate kernel's shared counter), de-duplication table size
ulimit -c 0
mkdir -p gpurun_out
cp krepp_amd/lib/libkrepp_amd.so /tmp/main_lib.so
for w in syn1000 syn10000; do
  bash scripts/ktimes.sh ${w}_main --workload $w > gpurun_out/r5_s7_ktimes_${w}_main.txt 2>&1
  echo "== $w main"; grep "acc_kernel_t<true, 5, false, 7\|dedup_kernel\|select\|scan_pipe\|clear" gpurun_out/r5_s7_ktimes_${w}_main.txt
  cp krepp_amd/lib/variants/rec2048/libkrepp_amd.so krepp_amd/lib/libkrepp_amd.so
  bash scripts/ktimes.sh ${w}_rec2048 --workload $w > gpurun_out/r5_s7_ktimes_${w}_rec2048.txt 2>&1
  echo "== $w rec2048"; grep "acc_kernel_t<true, 5, false, 7\|dedup_kernel\|select\|scan_pipe" gpurun_out/r5_s7_ktimes_${w}_rec2048.txt
  cp /tmp/main_lib.so krepp_amd/lib/libkrepp_amd.so
  for sh in 2 3; do
    KR_DD_SHIFT=$sh bash scripts/ktimes.sh ${w}_sh$sh --workload $w > gpurun_out/r5_s7_ktimes_${w}_sh$sh.txt 2>&1
    echo "== $w dd_shift $sh"; grep "dedup\|select\|llh" gpurun_out/r5_s7_ktimes_${w}_sh$sh.txt
  done
done

##################### session 8 #####################
# round 5, session 8: select kernel by groups of lanes (A/B), place counters on lines of their own + reads per visit (A/B)
ulimit -c 0
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_syn1000.py::test_syn1000_10gb_index_vs_oracle_and_full_batch_properties --deselect tests/test_gpu_bench.py > gpurun_out/r5_s8_tests.txt 2>&1
grep -n "passed\|failed" gpurun_out/r5_s8_tests.txt
cp krepp_amd/lib/libkrepp_amd.so /tmp/main_lib.so
for w in syn1000 syn10000; do
  bash scripts/ktimes.sh ${w}_main --workload $w > gpurun_out/r5_s8_ktimes_${w}_main.txt 2>&1
  echo "== $w main (select by groups)"; grep "select" gpurun_out/r5_s8_ktimes_${w}_main.txt
  cp krepp_amd/lib/variants/selgrp0/libkrepp_amd.so krepp_amd/lib/libkrepp_amd.so
  bash scripts/ktimes.sh ${w}_selgrp0 --workload $w > gpurun_out/r5_s8_ktimes_${w}_selgrp0.txt 2>&1
  echo "== $w one read at a time"; grep "select" gpurun_out/r5_s8_ktimes_${w}_selgrp0.txt
  cp /tmp/main_lib.so krepp_amd/lib/libkrepp_amd.so
done
python scripts/time_place_big.py > gpurun_out/r5_s8_place_main.txt 2>&1
echo "== place main"; cat gpurun_out/r5_s8_place_main.txt | cut -c1-220
cp krepp_amd/lib/variants/plrc16/libkrepp_amd.so krepp_amd/lib/libkrepp_amd.so
python scripts/time_place_big.py > gpurun_out/r5_s8_place_plrc16.txt 2>&1
echo "== place, 16 reads per visit"; cat gpurun_out/r5_s8_place_plrc16.txt | cut -c1-220
cp /tmp/main_lib.so krepp_amd/lib/libkrepp_amd.so

##################### session 9 #####################
# round 5, session 9: the whole GPU suite and smoke() at HEAD
ulimit -c 0
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q --durations=6 > gpurun_out/r5_s9_tests.txt 2>&1
grep -n "passed\|failed\|error" gpurun_out/r5_s9_tests.txt | head; tail -9 gpurun_out/r5_s9_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5_s9_smoke.txt 2>&1; tail -2 gpurun_out/r5_s9_smoke.txt

##################### session 10 #####################
# round 5, session 10: the profile set at HEAD (rocprofv3 stats + PMC on both indexes), the bench lines it must reproduce
# (default flags, the driver's flags, --workload syn10000), and the CLI end to end on both indexes
TAG=r5a
ulimit -c 0
mkdir -p gpurun_out
bash scripts/profile.sh $TAG > gpurun_out/${TAG}_profile.log 2>&1
python3 bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python3 scripts/traffic.py gpurun_out/prof_$TAG gpurun_out/${TAG}_bench.json gpurun_out/${TAG}_traffic.json > gpurun_out/${TAG}_traffic.log 2>&1
cp gpurun_out/${TAG}_traffic.json profiles/traffic_latest.json
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driver.json 2> gpurun_out/${TAG}_bench_driver.err
tail -c 400 gpurun_out/${TAG}_bench_driver.json
bash scripts/profile.sh ${TAG}_s10k --workload syn10000 > gpurun_out/${TAG}_s10k_profile.log 2>&1
python3 bench.py --workload syn10000 --no-cpu-baseline > gpurun_out/${TAG}_s10k_bench.json 2> gpurun_out/${TAG}_s10k_bench.err
tail -c 300 gpurun_out/${TAG}_s10k_bench.json
python scripts/time_cli.py 16000000 > gpurun_out/${TAG}_cli_toy25.txt 2>&1
KR_TIME_CLI_TRACE=1 python scripts/time_cli_syn1000.py 8e6 > gpurun_out/${TAG}_cli_syn1000.txt 2>&1
grep "rc 0\|kr_" gpurun_out/${TAG}_cli_syn1000.txt | cut -c1-200

##################### session 11 #####################
# round 5, session 11: fuzzers on the GPU (text against the host formatter; configurations, alphabet, long sequences against the
# oracle), the kernels of a CLI batch by name, place by phase
ulimit -c 0
mkdir -p gpurun_out
python scripts/fuzz_text.py 10 > gpurun_out/r5_s11_fuzz_text.txt 2>&1; tail -3 gpurun_out/r5_s11_fuzz_text.txt
python scripts/fuzz_reads.py 6 > gpurun_out/r5_s11_fuzz_reads.txt 2>&1; tail -2 gpurun_out/r5_s11_fuzz_reads.txt
python scripts/sweep_configs.py 11 > gpurun_out/r5_s11_sweep_configs.txt 2>&1; tail -2 gpurun_out/r5_s11_sweep_configs.txt
python scripts/fuzz_long.py 5 > gpurun_out/r5_s11_fuzz_long.txt 2>&1; tail -2 gpurun_out/r5_s11_fuzz_long.txt
python scripts/sweep_libs.py > gpurun_out/r5_s11_sweep_libs.txt 2>&1; tail -2 gpurun_out/r5_s11_sweep_libs.txt
KR_TIME_CLI_CONFIGS=0 KR_TIME_CLI_TRACE=1 python scripts/time_cli_syn1000.py 8e6 > gpurun_out/r5_s11_cli_trace.txt 2>&1
grep "rc 0\|kr_\|kernels of" gpurun_out/r5_s11_cli_trace.txt | cut -c1-200
KR_PLACE_TIMING=1 python scripts/time_place_big.py > gpurun_out/r5_s11_place_timing.txt 2>&1
grep "place/device\|tabular: 400000" gpurun_out/r5_s11_place_timing.txt | head -40 | cut -c1-200

##################### session 12 #####################
# round 5, session 12: the FASTQ parser with SSE2 record checks, chunks parsed from a mapping of the file -- CLI end to end again
ulimit -c 0
mkdir -p gpurun_out
python -m pytest tests/test_place.py tests/test_gpu_rccl_cli.py tests/test_gpu_text.py -m gpu -x -q > gpurun_out/r5_s12_tests.txt 2>&1
grep -n "passed\|failed" gpurun_out/r5_s12_tests.txt
python scripts/time_gz.py > gpurun_out/r5_s12_reader.txt 2>&1; tail -25 gpurun_out/r5_s12_reader.txt | cut -c1-200
python scripts/time_cli.py 16000000 > gpurun_out/r5_s12_cli_toy25.txt 2>&1
grep "elapsed" gpurun_out/r5_s12_cli_toy25.txt | grep -o "^[a-z]* \[[^]]*\] {[^}]*}\|elapsed: [0-9.]* sec ([0-9]* reads/s" | paste - - | head -20
KR_TIME_CLI_CONFIGS=0,6,7,8 python scripts/time_cli_syn1000.py 8e6 > gpurun_out/r5_s12_cli_syn1000.txt 2>&1
grep "rc 0\|parse" gpurun_out/r5_s12_cli_syn1000.txt | cut -c1-220
KR_FASTX_MMAP=0 KR_TIME_CLI_CONFIGS=8 python scripts/time_cli_syn1000.py 8e6 2>&1 | grep "rc 0\|parse" | cut -c1-220

##################### session 13 #####################
# round 5, session 13: kr_place_stream by ranges of reads -- parity (every place test) and rate
ulimit -c 0
mkdir -p gpurun_out
python -m pytest tests/test_place.py tests/test_gpu_place_k27.py "tests/test_gpu_syn1000.py::test_place_on_the_1000_genome_tree_never_leaves_the_device" "tests/test_gpu_syn1000.py::test_place_on_the_1000_genome_tree_matches_the_oracle" -m gpu -x -q > gpurun_out/r5_s13_tests.txt 2>&1
grep -n "passed\|failed" gpurun_out/r5_s13_tests.txt; tail -5 gpurun_out/r5_s13_tests.txt | cut -c1-200
python scripts/sweep_place.py > gpurun_out/r5_s13_sweep_place.txt 2>&1; tail -2 gpurun_out/r5_s13_sweep_place.txt
for k in 4 1 2 8; do
  echo "== KR_PLACE_RANGES=$k"
  KR_PLACE_RANGES=$k python scripts/time_place_big.py > gpurun_out/r5_s13_place_ranges$k.txt 2>&1
  cut -c1-200 gpurun_out/r5_s13_place_ranges$k.txt | head -6
done
python scripts/time_cli.py 16000000 2>&1 | grep "^place" | grep -o "^place [^{]*\|elapsed: [0-9.]* sec ([0-9]* reads/s" | paste - - | head
#!/bin/bash
# round 5, session 14: the whole GPU suite, smoke() and the driver's bench command at HEAD
ulimit -c 0
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q --durations=5 > gpurun_out/r5_s14_tests.txt 2>&1
grep -n "passed\|failed\|error" gpurun_out/r5_s14_tests.txt | head -5
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5_s14_bench_driver.json 2> gpurun_out/r5_s14_bench_driver.err
python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/r5_s14_bench_driver.json') if l.startswith('{')][-1]); h=d['value_host_inclusive']
print(round(d['value']/1e6,2), round(d['ms_per_step'],2), d['kernel_ms']['scan'], d['kernel_ms']['accumulate'], d['kernel_ms']['llh_select'], 'frac', round(d['roofline']['frac'],3), d['roofline']['frac_traffic'], 'host', round(h['value']/1e6,2), round(h['steady_state']['value']/1e6,2), d['check']['rows_equal'], d['cpu_baseline']['value'])"
python scripts/time_cli.py 16000000 2>&1 | grep "elapsed" | grep -o "^[a-z]* \[[^]]*\] {[^}]*}\|elapsed: [0-9.]* sec ([0-9]* reads/s" | paste - - | head -12
KR_TIME_CLI_CONFIGS=0,6,7,8 python scripts/time_cli_syn1000.py 8e6 2>&1 | grep "rc 0" | cut -c1-200
# round 5, session 15: the straight-line epilogue with counts instead of planes -- parity, then A/B by kernel time on both indexes
ulimit -c 0
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_text.py tests/test_gpu_long_sequences.py tests/test_gpu_filter_slots.py tests/test_gpu_syn1000.py::test_ten_thousand_genome_index_vs_oracle -x -q > gpurun_out/r5_s15_tests.txt 2>&1
grep -n "passed\|failed" gpurun_out/r5_s15_tests.txt; tail -3 gpurun_out/r5_s15_tests.txt | cut -c1-200
cp krepp_amd/lib/libkrepp_amd.so /tmp/main_lib.so
for w in syn1000 syn10000; do
  bash scripts/ktimes.sh ${w}_main --workload $w > gpurun_out/r5_s15_ktimes_${w}_counts.txt 2>&1
  echo "== $w counts"; grep "acc_kernel_t<true, 5, false, 7\|scan_pipe" gpurun_out/r5_s15_ktimes_${w}_counts.txt
  cp krepp_amd/lib/variants/planes/libkrepp_amd.so krepp_amd/lib/libkrepp_amd.so
  bash scripts/ktimes.sh ${w}_planes --workload $w > gpurun_out/r5_s15_ktimes_${w}_planes.txt 2>&1
  echo "== $w planes"; grep "acc_kernel_t<true, 5, false, 7\|scan_pipe" gpurun_out/r5_s15_ktimes_${w}_planes.txt
  cp /tmp/main_lib.so krepp_amd/lib/libkrepp_amd.so
done
cp krepp_amd/lib/variants/stats/libkrepp_amd.so krepp_amd/lib/libkrepp_amd.so
S="--no-cpu-baseline --no-host-inclusive --steps 1 --warmup 0 --check-reads 1000 --skip-host-path-check --distinct-batches 1"
for w in syn1000 syn10000; do KR_ITEM_PLACEMENT_TRIALS=0 KR_DEBUG_SKIP=512 python bench.py --workload $w $S 2>&1 | grep "kr stats\] paths\|kr stats\] reads [0-9]" | cut -c1-330; done
cp /tmp/main_lib.so krepp_amd/lib/libkrepp_amd.so
# round 5, session 16: HEAD after the count epilogue -- the whole GPU suite, the profile set again (stage digests), bench lines
TAG=r5b
ulimit -c 0
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_tests.txt 2>&1
grep -n "passed\|failed\|error" gpurun_out/${TAG}_tests.txt | head -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash scripts/profile.sh $TAG > gpurun_out/${TAG}_profile.log 2>&1
python3 bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python3 scripts/traffic.py gpurun_out/prof_$TAG gpurun_out/${TAG}_bench.json gpurun_out/${TAG}_traffic.json > gpurun_out/${TAG}_traffic.log 2>&1
cp gpurun_out/${TAG}_traffic.json profiles/traffic_latest.json
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driver.json 2> gpurun_out/${TAG}_bench_driver.err
tail -c 300 gpurun_out/${TAG}_bench_driver.json
python3 bench.py --workload syn10000 --no-cpu-baseline > gpurun_out/${TAG}_s10k_bench.json 2> gpurun_out/${TAG}_s10k_bench.err
tail -c 200 gpurun_out/${TAG}_s10k_bench.json
# round 5, session 17: where the straight-line epilogue's time is -- the accumulate kernel with the epilogue cut short at four points
# (scripts/acc_epilogue_ablation.patch builds the variants: results wrong by design, launch times are what is read)
ulimit -c 0
mkdir -p gpurun_out
cp krepp_amd/lib/libkrepp_amd.so /tmp/main_lib.so
for w in syn1000 syn10000; do
  bash scripts/ktimes.sh ${w}_full --workload $w > gpurun_out/r5_s17_${w}_full.txt 2>&1
  echo "== $w whole kernel"; grep "acc_kernel_t<true, 5, false, 7" gpurun_out/r5_s17_${w}_full.txt
  for a in 1 2 3 4; do
    cp krepp_amd/lib/variants/abl$a/libkrepp_amd.so krepp_amd/lib/libkrepp_amd.so
    bash scripts/ktimes.sh ${w}_abl$a --workload $w > gpurun_out/r5_s17_${w}_abl$a.txt 2>&1
    echo "== $w cut after step $a"; grep "acc_kernel_t<true, 5, false, 7" gpurun_out/r5_s17_${w}_abl$a.txt
  done
  cp /tmp/main_lib.so krepp_amd/lib/libkrepp_amd.so
done
# a traced process on the scan's fast level, for the kernel-stats summary at HEAD (up to four tries)
for t in 1 2 3 4; do
  bash scripts/profile_trace_only.sh r5c$t > gpurun_out/r5c${t}_trace.log 2>&1
  tail -1 gpurun_out/r5c${t}_trace.log
done
# round 5, session 18: straight-line epilogue for up to 128 keys (two rounds of the per-key step) -- parity, A/B against 64, path statistics
ulimit -c 0
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_text.py tests/test_gpu_long_sequences.py tests/test_gpu_filter_slots.py tests/test_gpu_syn1000.py::test_ten_thousand_genome_index_vs_oracle tests/test_gpu_place_k27.py -x -q > gpurun_out/r5_s18_tests.txt 2>&1
grep -n "passed\|failed" gpurun_out/r5_s18_tests.txt; tail -3 gpurun_out/r5_s18_tests.txt | cut -c1-200
cp krepp_amd/lib/libkrepp_amd.so /tmp/main_lib.so
for w in syn1000 syn10000; do
  bash scripts/ktimes.sh ${w}_k128 --workload $w > gpurun_out/r5_s18_ktimes_${w}_k128.txt 2>&1
  echo "== $w 128 keys"; grep "acc_kernel_t<true, 5, false, 7" gpurun_out/r5_s18_ktimes_${w}_k128.txt
  cp krepp_amd/lib/variants/keys64/libkrepp_amd.so krepp_amd/lib/libkrepp_amd.so
  bash scripts/ktimes.sh ${w}_k64 --workload $w > gpurun_out/r5_s18_ktimes_${w}_k64.txt 2>&1
  echo "== $w 64 keys"; grep "acc_kernel_t<true, 5, false, 7" gpurun_out/r5_s18_ktimes_${w}_k64.txt
  cp /tmp/main_lib.so krepp_amd/lib/libkrepp_amd.so
done
cp krepp_amd/lib/variants/stats/libkrepp_amd.so krepp_amd/lib/libkrepp_amd.so
S="--no-cpu-baseline --no-host-inclusive --steps 1 --warmup 0 --check-reads 1000 --skip-host-path-check --distinct-batches 1"
for w in syn1000 syn10000; do KR_ITEM_PLACEMENT_TRIALS=0 KR_DEBUG_SKIP=512 python bench.py --workload $w $S 2>&1 | grep "kr stats\] paths" | cut -c1-330; done
cp /tmp/main_lib.so krepp_amd/lib/libkrepp_amd.so
# round 5, session 19: what the GENERAL epilogue costs, and for which reads (scripts/acc_general_epilogue_ablation.patch: 5 = skipped for every
# read the straight-line epilogue turns away, 6 = skipped for those whose events spilled out of the LDS, 7 = skipped for the others)
ulimit -c 0
mkdir -p gpurun_out
cp krepp_amd/lib/libkrepp_amd.so /tmp/main_lib.so
for w in syn1000 syn10000; do
  bash scripts/ktimes.sh ${w}_full --workload $w > gpurun_out/r5_s19_${w}_full.txt 2>&1
  echo "== $w whole kernel"; grep "acc_kernel_t<true, 5, false, 7" gpurun_out/r5_s19_${w}_full.txt
  for a in 5 6 7; do
    cp krepp_amd/lib/variants/abl$a/libkrepp_amd.so krepp_amd/lib/libkrepp_amd.so
    bash scripts/ktimes.sh ${w}_abl$a --workload $w > gpurun_out/r5_s19_${w}_abl$a.txt 2>&1
    echo "== $w general epilogue skipped, mode $a"; grep "acc_kernel_t<true, 5, false, 7" gpurun_out/r5_s19_${w}_abl$a.txt
  done
  cp /tmp/main_lib.so krepp_amd/lib/libkrepp_amd.so
done
# round 5, session 20: events of spilled reads compacted into the LDS (KR_ACC_COMPACT_SPILLED): parity, then the accumulate kernel's time
# with and without on both indexes
ulimit -c 0
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_text.py tests/test_gpu_long_sequences.py tests/test_gpu_filter_slots.py \
  tests/test_gpu_syn1000.py tests/test_gpu_place_k27.py -x -q -m gpu > gpurun_out/r5_s20_tests.txt 2>&1
tail -3 gpurun_out/r5_s20_tests.txt
cp krepp_amd/lib/libkrepp_amd.so /tmp/main_lib.so
for w in syn1000 syn10000; do
  bash scripts/ktimes.sh ${w}_compact --workload $w > gpurun_out/r5_s20_${w}_compact.txt 2>&1
  echo "== $w compacted"; grep "acc_kernel_t<true, 5, false, 7\|sum of max" gpurun_out/r5_s20_${w}_compact.txt
  cp krepp_amd/lib/variants/nocompact/libkrepp_amd.so krepp_amd/lib/libkrepp_amd.so
  bash scripts/ktimes.sh ${w}_nocompact --workload $w > gpurun_out/r5_s20_${w}_nocompact.txt 2>&1
  echo "== $w not compacted"; grep "acc_kernel_t<true, 5, false, 7\|sum of max" gpurun_out/r5_s20_${w}_nocompact.txt
  cp /tmp/main_lib.so krepp_amd/lib/libkrepp_amd.so
done
python bench.py > gpurun_out/r5_s20_bench.json 2> gpurun_out/r5_s20_bench.err; cat gpurun_out/r5_s20_bench.json | cut -c1-400
# round 5, session 21: straight-line epilogue with key batches and exact handling of a position hit twice, spilled reads compacted up to
# 768 live events: parity (whole suite), then the accumulate kernel's time with and without the compaction
ulimit -c 0
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_long_sequences.py tests/test_gpu_syn1000.py -x -q -m gpu > gpurun_out/r5_s21_tests.txt 2>&1
tail -3 gpurun_out/r5_s21_tests.txt
cp krepp_amd/lib/libkrepp_amd.so /tmp/main_lib.so
for w in syn1000 syn10000; do
  bash scripts/ktimes.sh ${w}_compact --workload $w > gpurun_out/r5_s21_${w}_compact.txt 2>&1
  echo "== $w compacted"; grep "acc_kernel_t<true, 5, false, 7\|sum of max" gpurun_out/r5_s21_${w}_compact.txt
  cp krepp_amd/lib/variants/nocompact/libkrepp_amd.so krepp_amd/lib/libkrepp_amd.so
  bash scripts/ktimes.sh ${w}_nocompact --workload $w > gpurun_out/r5_s21_${w}_nocompact.txt 2>&1
  echo "== $w not compacted"; grep "acc_kernel_t<true, 5, false, 7\|sum of max" gpurun_out/r5_s21_${w}_nocompact.txt
  cp /tmp/main_lib.so krepp_amd/lib/libkrepp_amd.so
done
python bench.py > gpurun_out/r5_s21_bench.json 2> gpurun_out/r5_s21_bench.err; cut -c1-300 gpurun_out/r5_s21_bench.json
python bench.py --workload syn10000 --no-cpu-baseline > gpurun_out/r5_s21_bench_syn10000.json 2> gpurun_out/r5_s21_bench_syn10000.err; cut -c1-300 gpurun_out/r5_s21_bench_syn10000.json
timeout 1200 python -m pytest tests/ -x -q -m gpu --deselect tests/test_gpu_parity.py --deselect tests/test_gpu_long_sequences.py --deselect tests/test_gpu_syn1000.py > gpurun_out/r5_s21_tests2.txt 2>&1
tail -3 gpurun_out/r5_s21_tests2.txt
# round 5, session 22: in-place compaction for reads whose keys do not fit one batch; the LDS event capacity at 768 and 1024 instead of 512;
# what the reads with spilled events still cost (scripts/acc_spilled_reads_ablation.diff: 8 = such a read does nothing after its events
# are collected, 9 = nothing after the compaction)
ulimit -c 0
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_syn1000.py::test_ten_thousand_genome_index_vs_oracle -x -q -m gpu > gpurun_out/r5_s22_tests.txt 2>&1
tail -3 gpurun_out/r5_s22_tests.txt
cp krepp_amd/lib/libkrepp_amd.so /tmp/main_lib.so
for w in syn1000 syn10000; do
  for v in main ev768 ev1024 abl8 abl9; do
    if [ $w = syn10000 ] && [ ${v#abl} != $v ]; then continue; fi
    if [ $v = main ]; then cp /tmp/main_lib.so krepp_amd/lib/libkrepp_amd.so; else cp krepp_amd/lib/variants/$v/libkrepp_amd.so krepp_amd/lib/libkrepp_amd.so; fi
    bash scripts/ktimes.sh ${w}_$v --workload $w > gpurun_out/r5_s22_${w}_$v.txt 2>&1
    echo "== $w $v"; grep "acc_kernel_t<true, 5, false, 7\|sum of max" gpurun_out/r5_s22_${w}_$v.txt
  done
done
cp /tmp/main_lib.so krepp_amd/lib/libkrepp_amd.so
# round 5, session 23: 1,024 events in the LDS, one compaction pass (what does not fit goes back to the global scratch), spilled tiles loaded
# four at a time; parity, then what is left (scripts/acc_spilled_reads_ablation.diff: 8 = a read with spilled events does nothing after its
# events are collected, 9 = nothing after the compaction, 10 = no read does anything after marks / compaction)
ulimit -c 0
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_syn1000.py tests/test_gpu_long_sequences.py -x -q -m gpu > gpurun_out/r5_s23_tests.txt 2>&1
tail -3 gpurun_out/r5_s23_tests.txt
cp krepp_amd/lib/libkrepp_amd.so /tmp/main_lib.so
for w in syn1000 syn10000; do
  for v in main abl8 abl9 abl10; do
    if [ $v = main ]; then cp /tmp/main_lib.so krepp_amd/lib/libkrepp_amd.so; else cp krepp_amd/lib/variants/$v/libkrepp_amd.so krepp_amd/lib/libkrepp_amd.so; fi
    bash scripts/ktimes.sh ${w}_$v --workload $w > gpurun_out/r5_s23_${w}_$v.txt 2>&1
    echo "== $w $v"; grep "acc_kernel_t<true, 5, false, 7\|sum of max" gpurun_out/r5_s23_${w}_$v.txt
  done
done
cp /tmp/main_lib.so krepp_amd/lib/libkrepp_amd.so
# round 5, session 24: HEAD after the epilogue work of sessions 20-23 -- the whole GPU suite, smoke, the profile set, bench lines
TAG=r5c
ulimit -c 0
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_tests.txt 2>&1
grep -n "passed\|failed\|error" gpurun_out/${TAG}_tests.txt | head -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash scripts/profile.sh $TAG > gpurun_out/${TAG}_profile.log 2>&1
python3 bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python3 scripts/traffic.py gpurun_out/prof_$TAG gpurun_out/${TAG}_bench.json gpurun_out/${TAG}_traffic.json > gpurun_out/${TAG}_traffic.log 2>&1
cp gpurun_out/${TAG}_traffic.json profiles/traffic_latest.json
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driver.json 2> gpurun_out/${TAG}_bench_driver.err
tail -c 300 gpurun_out/${TAG}_bench_driver.json
python3 bench.py --workload syn10000 --no-cpu-baseline > gpurun_out/${TAG}_s10k_bench.json 2> gpurun_out/${TAG}_s10k_bench.err
tail -c 200 gpurun_out/${TAG}_s10k_bench.json
# round 5, session 25: reads whose live events do not fit the LDS finished from position maps and counters in the wave's global scratch
# (finish_big_read) instead of finalize_events' global planes: parity (also with every key of such a read counted again from the events:
# the path a position hit twice takes), then the accumulate kernel with and without
ulimit -c 0
mkdir -p gpurun_out
BIG="tests/test_gpu_parity.py::test_large_clade_colours_spill_the_work_stack tests/test_gpu_parity.py::test_forty_thousand_leaves tests/test_gpu_parity.py::test_many_leaves_bitmap_spans_several_tiles tests/test_gpu_parity.py::test_single_segment_many_leaves_spill_paths tests/test_gpu_parity.py::test_overflow_path_many_leaves tests/test_gpu_parity.py::test_crafted_min_rule_null_nodes_and_th tests/test_gpu_syn1000.py::test_ten_thousand_genome_index_vs_oracle"
cp krepp_amd/lib/libkrepp_amd.so /tmp/main_lib.so
cp krepp_amd/lib/variants/recount/libkrepp_amd.so krepp_amd/lib/libkrepp_amd.so
timeout 900 python -m pytest $BIG -x -q -m gpu > gpurun_out/r5_s25_tests_recount.txt 2>&1
echo "== every key counted again:"; tail -3 gpurun_out/r5_s25_tests_recount.txt | cut -c1-300
cp /tmp/main_lib.so krepp_amd/lib/libkrepp_amd.so
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_syn1000.py tests/test_gpu_long_sequences.py -x -q -m gpu > gpurun_out/r5_s25_tests.txt 2>&1
tail -3 gpurun_out/r5_s25_tests.txt | cut -c1-300
for w in syn10000 syn1000; do
  for v in main nobig; do
    if [ $v = main ]; then cp /tmp/main_lib.so krepp_amd/lib/libkrepp_amd.so; else cp krepp_amd/lib/variants/$v/libkrepp_amd.so krepp_amd/lib/libkrepp_amd.so; fi
    bash scripts/ktimes.sh ${w}_$v --workload $w > gpurun_out/r5_s25_${w}_$v.txt 2>&1
    echo "== $w $v"; grep "acc_kernel_t<true, 5, false, 7\|sum of max" gpurun_out/r5_s25_${w}_$v.txt
  done
done
cp /tmp/main_lib.so krepp_amd/lib/libkrepp_amd.so
# round 5, session 26: finish_big_read as a function of its own (inlined it cost the 1000-genome index 2.8 ms: session 25) -- the tests with
# big reads (also with every key counted again), the accumulate kernel with and without
ulimit -c 0
mkdir -p gpurun_out
BIG="tests/test_gpu_parity.py::test_large_clade_colours_spill_the_work_stack tests/test_gpu_parity.py::test_forty_thousand_leaves tests/test_gpu_parity.py::test_many_leaves_bitmap_spans_several_tiles tests/test_gpu_parity.py::test_single_segment_many_leaves_spill_paths tests/test_gpu_parity.py::test_overflow_path_many_leaves tests/test_gpu_parity.py::test_crafted_min_rule_null_nodes_and_th tests/test_gpu_syn1000.py::test_ten_thousand_genome_index_vs_oracle"
cp krepp_amd/lib/libkrepp_amd.so /tmp/main_lib.so
cp krepp_amd/lib/variants/recount/libkrepp_amd.so krepp_amd/lib/libkrepp_amd.so
timeout 900 python -m pytest $BIG -x -q -m gpu > gpurun_out/r5_s26_tests_recount.txt 2>&1
echo "== every key counted again:"; tail -3 gpurun_out/r5_s26_tests_recount.txt | cut -c1-300
cp /tmp/main_lib.so krepp_amd/lib/libkrepp_amd.so
timeout 900 python -m pytest $BIG -x -q -m gpu > gpurun_out/r5_s26_tests.txt 2>&1
tail -3 gpurun_out/r5_s26_tests.txt | cut -c1-300
for w in syn1000 syn10000; do
  for v in main nobig; do
    if [ $v = main ]; then cp /tmp/main_lib.so krepp_amd/lib/libkrepp_amd.so; else cp krepp_amd/lib/variants/$v/libkrepp_amd.so krepp_amd/lib/libkrepp_amd.so; fi
    bash scripts/ktimes.sh ${w}_$v --workload $w > gpurun_out/r5_s26_${w}_$v.txt 2>&1
    echo "== $w $v"; grep "acc_kernel_t<true, 5, false, 7\|sum of max" gpurun_out/r5_s26_${w}_$v.txt
  done
done
cp /tmp/main_lib.so krepp_amd/lib/libkrepp_amd.so
# round 5, session 27: finish_big_read only for reads whose live events overflow the LDS (session 26: the reads that merely fill it are
# faster in finalize_events) -- the tests with big reads, the accumulate kernel on both indexes
ulimit -c 0
mkdir -p gpurun_out
BIG="tests/test_gpu_parity.py::test_large_clade_colours_spill_the_work_stack tests/test_gpu_parity.py::test_forty_thousand_leaves tests/test_gpu_parity.py::test_many_leaves_bitmap_spans_several_tiles tests/test_gpu_parity.py::test_single_segment_many_leaves_spill_paths tests/test_gpu_parity.py::test_crafted_min_rule_null_nodes_and_th tests/test_gpu_syn1000.py::test_ten_thousand_genome_index_vs_oracle"
timeout 900 python -m pytest $BIG -x -q -m gpu > gpurun_out/r5_s27_tests.txt 2>&1
tail -3 gpurun_out/r5_s27_tests.txt | cut -c1-300
for w in syn1000 syn10000; do
  bash scripts/ktimes.sh ${w}_main --workload $w > gpurun_out/r5_s27_${w}_main.txt 2>&1
  echo "== $w main"; grep "acc_kernel_t<true, 5, false, 7\|sum of max" gpurun_out/r5_s27_${w}_main.txt
done
# round 5, session 28: finish_big_read with planes in the global scratch and no returning atomic -- the tests with big reads, the accumulate
# kernel on both indexes with and without
ulimit -c 0
mkdir -p gpurun_out
BIG="tests/test_gpu_parity.py::test_large_clade_colours_spill_the_work_stack tests/test_gpu_parity.py::test_forty_thousand_leaves tests/test_gpu_parity.py::test_many_leaves_bitmap_spans_several_tiles tests/test_gpu_parity.py::test_single_segment_many_leaves_spill_paths tests/test_gpu_parity.py::test_crafted_min_rule_null_nodes_and_th tests/test_gpu_syn1000.py::test_ten_thousand_genome_index_vs_oracle"
timeout 900 python -m pytest $BIG -x -q -m gpu > gpurun_out/r5_s28_tests.txt 2>&1
tail -3 gpurun_out/r5_s28_tests.txt | cut -c1-300
cp krepp_amd/lib/libkrepp_amd.so /tmp/main_lib.so
for w in syn1000 syn10000; do
  for v in main nobig; do
    if [ $v = main ]; then cp /tmp/main_lib.so krepp_amd/lib/libkrepp_amd.so; else cp krepp_amd/lib/variants/$v/libkrepp_amd.so krepp_amd/lib/libkrepp_amd.so; fi
    bash scripts/ktimes.sh ${w}_$v --workload $w > gpurun_out/r5_s28_${w}_$v.txt 2>&1
    echo "== $w $v"; grep "acc_kernel_t<true, 5, false, 7\|sum of max" gpurun_out/r5_s28_${w}_$v.txt
  done
done
cp /tmp/main_lib.so krepp_amd/lib/libkrepp_amd.so
# round 5, session 29: HEAD at the end of the round -- the whole GPU suite, smoke, the profile set (digests current), bench lines
TAG=r5d
ulimit -c 0
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_tests.txt 2>&1
grep -n "passed\|failed\|error" gpurun_out/${TAG}_tests.txt | head -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash scripts/profile.sh $TAG > gpurun_out/${TAG}_profile.log 2>&1
python3 bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python3 scripts/traffic.py gpurun_out/prof_$TAG gpurun_out/${TAG}_bench.json gpurun_out/${TAG}_traffic.json > gpurun_out/${TAG}_traffic.log 2>&1
cp gpurun_out/${TAG}_traffic.json profiles/traffic_latest.json
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driver.json 2> gpurun_out/${TAG}_bench_driver.err
tail -c 300 gpurun_out/${TAG}_bench_driver.json
python3 bench.py --workload syn10000 --no-cpu-baseline > gpurun_out/${TAG}_s10k_bench.json 2> gpurun_out/${TAG}_s10k_bench.err
tail -c 200 gpurun_out/${TAG}_s10k_bench.json
# round 5, sessions 30-32: the CLI on the benchmark index at HEAD (30: output path reused, 31: a fresh output file per run), then the profile
# passes once more at the final sources (32)
KR_TIME_CLI_CONFIGS=0,7,8,0,7,8 timeout 400 python scripts/time_cli_syn1000.py 8e6 > gpurun_out/r5_s30_cli_syn1000.txt 2>&1
KR_TIME_CLI_CONFIGS=0,2,3,4,1,7,8,0 timeout 400 python scripts/time_cli_syn1000.py 8e6 > gpurun_out/r5_s31_cli_syn1000.txt 2>&1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash scripts/profile.sh r5e > gpurun_out/r5e_profile.log 2>&1
python3 scripts/traffic.py gpurun_out/prof_r5e gpurun_out/prof_r5e/bench_trace.log gpurun_out/r5e_traffic.json > gpurun_out/r5e_traffic.log 2>&1
# round 5, session 33: --no-multi / --summarize with three and four workers, 524,288-read batches
KR_TIME_CLI_CONFIGS=8,9,10,11,12,8,9 timeout 190 python scripts/time_cli_syn1000.py 8e6 > gpurun_out/r5_s33_cli_syn1000.txt 2>&1
