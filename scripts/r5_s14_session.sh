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
