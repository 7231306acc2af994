#!/bin/bash
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
