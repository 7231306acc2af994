#!/bin/bash
# round 6, session 41: the de-duplication kernel as a software pipeline (loads two records ahead, the first look at the table slot
# one record ahead) -- HEAD against -DKR_DEDUP_PIPE=0 (variants/dpipe0); per-kernel times by rocprofv3, then the stage by bench.py
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/s41
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8
cp krepp_amd/lib/libkrepp_amd.so /tmp/base.so
for wl in syn1000 syn10000; do
for v in dpipe0 base; do
  if [ $v = base ]; then cp /tmp/base.so krepp_amd/lib/libkrepp_amd.so; else cp krepp_amd/lib/variants/$v/libkrepp_amd.so krepp_amd/lib/libkrepp_amd.so; fi
  timeout 600 python bench.py --workload $wl --steps 8 --warmup 3 --no-cpu-baseline --no-host-inclusive > gpurun_out/s41/${wl}_$v.json 2> gpurun_out/s41/${wl}_$v.err
  python - <<PY
import json
try:
    o=json.loads([l for l in open("gpurun_out/s41/${wl}_$v.json") if l.startswith("{")][0])
    k={x["stage"]:round(x["avg_launch_ms"],2) for x in o["roofline"]["kernels"]}
    print("$wl $v", round(o["value"]/1e6,2), k, o["check"]["rows_equal"], o["check"]["whole_launch"]["equal_on_an_independent_stream"])
except Exception as e: print("$wl $v failed", e)
PY
done; done
cp /tmp/base.so krepp_amd/lib/libkrepp_amd.so
bash scripts/r6_s40_kernel_times.sh 2>&1 | grep -E "==|dedup_kernel|select_lane|llh_kernel"
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_syn1000.py -x -q -m gpu > gpurun_out/s41/pytest.txt 2>&1; grep -E "passed|failed|error" gpurun_out/s41/pytest.txt | tail -n 3
