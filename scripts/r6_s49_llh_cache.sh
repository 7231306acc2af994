#!/bin/bash
# round 6, session 49: the table of likelihood problems kept across batches (opt-in, KR_LLH_CACHE_LOG2): its own test, the parity / place /
# text tests with it switched on for every stream, and bench.py with and without it -- eight distinct batches, four of them warm-up, the four
# timed ones unseen (the default two cycled batches would find everything in the table)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/s49
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "kept_across" 2>&1 | tail -n 3
KR_LLH_CACHE_LOG2=22 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_place.py tests/test_gpu_text.py -x -q -m gpu -k "not kept_across" > gpurun_out/s49/pytest_cache_on.txt 2>&1; grep -E "passed|failed|error" gpurun_out/s49/pytest_cache_on.txt | tail -n 3
for wl in syn1000 syn10000; do
for lg in 0 26; do
  KR_LLH_CACHE_LOG2=$lg timeout 600 python bench.py --workload $wl --distinct-batches 8 --steps 4 --warmup 4 --no-cpu-baseline --no-host-inclusive > gpurun_out/s49/${wl}_$lg.json 2> gpurun_out/s49/${wl}_$lg.err
  python - <<PY
import json
try:
    o=json.loads([l for l in open("gpurun_out/s49/${wl}_$lg.json") if l.startswith("{")][0])
    k={x["stage"]:round(x["avg_launch_ms"],2) for x in o["roofline"]["kernels"]}
    print("$wl cache_log2=$lg", round(o["value"]/1e6,2), k, o["check"]["rows_equal"], o["check"]["whole_launch"]["equal_on_an_independent_stream"])
except Exception as e: print("$wl $lg failed", e)
PY
done; done
