#!/bin/bash
# round 6, session 35: the direct part's values beside its places (dd_dv: one look-up instead of two for four records in five) --
# base = HEAD, ddv_wpe6 = HEAD with -DKR_SELECT_WPE=6 (80 registers, six waves per SIMD in the select kernel)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/s35
cp krepp_amd/lib/libkrepp_amd.so /tmp/base.so
for wl in syn1000 syn10000; do
for v in base ddv_wpe6; do
  if [ $v = base ]; then cp /tmp/base.so krepp_amd/lib/libkrepp_amd.so; else cp krepp_amd/lib/variants/$v/libkrepp_amd.so krepp_amd/lib/libkrepp_amd.so; fi
  timeout 600 python bench.py --workload $wl --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/s35/${wl}_$v.json 2> gpurun_out/s35/${wl}_$v.err
  python - <<PY
import json
try:
    o=json.loads([l for l in open("gpurun_out/s35/${wl}_$v.json") if l.startswith("{")][0])
    k={x["stage"]:round(x["avg_launch_ms"],2) for x in o["roofline"]["kernels"]}
    print("$wl $v", round(o["value"]/1e6,2), k, o["check"]["rows_equal"], o["check"]["whole_launch"]["equal_on_an_independent_stream"], "host", round(o["value_host_inclusive"]["value"]/1e6,2), o["value_host_inclusive"].get("rows_equal"))
except Exception as e: print("$wl $v failed", e)
PY
done; done
cp /tmp/base.so krepp_amd/lib/libkrepp_amd.so
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_syn1000.py tests/test_gpu_text.py -x -q -m gpu > gpurun_out/s35/pytest.txt 2>&1; tail -n 3 gpurun_out/s35/pytest.txt
