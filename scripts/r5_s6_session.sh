#!/bin/bash
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
