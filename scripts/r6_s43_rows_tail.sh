#!/bin/bash
# round 6, session 43: the row kernel's reads of more than 64 records, four steps' loads at a time -- HEAD against the build before
# (variants/rows0); the host-inclusive leg of bench.py
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/s43
cp krepp_amd/lib/libkrepp_amd.so /tmp/base.so
for v in rows0 base rows0 base; do
  if [ $v = base ]; then cp /tmp/base.so krepp_amd/lib/libkrepp_amd.so; else cp krepp_amd/lib/variants/$v/libkrepp_amd.so krepp_amd/lib/libkrepp_amd.so; fi
  timeout 600 python bench.py --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/s43/$v.json 2> gpurun_out/s43/$v.err
  python - <<PY
import json
try:
    o=json.loads([l for l in open("gpurun_out/s43/$v.json") if l.startswith("{")][0])
    k={x["stage"]:round(x["avg_launch_ms"],2) for x in o["roofline"]["kernels"]}
    hi=o["value_host_inclusive"]
    print("$v", round(o["value"]/1e6,2), k, o["check"]["rows_equal"], "host", round(hi["value"]/1e6,2), "steady", round(hi["steady_state"]["value"]/1e6,2), hi.get("kernel_ms_in_this_leg"), hi.get("rows_equal"))
except Exception as e: print("$v failed", e)
PY
done
cp /tmp/base.so krepp_amd/lib/libkrepp_amd.so
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_syn1000.py -x -q -m gpu > gpurun_out/s43/pytest.txt 2>&1; grep -E "passed|failed|error" gpurun_out/s43/pytest.txt | tail -n 3
