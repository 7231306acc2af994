#!/bin/bash
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
