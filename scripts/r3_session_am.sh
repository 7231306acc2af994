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
