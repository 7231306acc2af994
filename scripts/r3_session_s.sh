#!/bin/bash
# Round 3, session S: do the scan's launch-time levels follow the GPU's clock levels?  (a) what the driver publishes, sampled
# next to a bench run whose streams are run again after idle gaps; (b) rocm-smi's view before and after.
ulimit -c 0
cd "$(dirname "$0")/.."
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out
ls -la /sys/class/drm/ 2>&1 | head -20
ls /sys/class/drm/card*/device/ 2>&1 | head -80
rocm-smi --showclocks --showpower --showperflevel --showtemp 2>&1 | head -60
python3 scripts/clock_probe.py 0.02 python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-host-inclusive \
   --stream-variance 5 --stream-variance-idle 1.0 > gpurun_out/r3s_clocks.txt 2> gpurun_out/r3s_bench.err
grep "stream-variance" gpurun_out/r3s_bench.err
head -5 gpurun_out/r3s_clocks.txt; wc -l gpurun_out/r3s_clocks.txt
# rocm-smi next to a second run (in case sysfs is not readable): one sample every ~0.3 s
python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-host-inclusive --stream-variance 4 --stream-variance-idle 1.0 \
   > gpurun_out/r3s_bench2.json 2> gpurun_out/r3s_bench2.err &
BP=$!
while kill -0 $BP 2>/dev/null; do
  echo "t=$(date +%s.%N)"; rocm-smi --showclocks --showpower 2>&1 | grep -E "clk|Power|power" ; sleep 0.1
done > gpurun_out/r3s_smi.txt
wait $BP
grep "stream-variance" gpurun_out/r3s_bench2.err
tail -40 gpurun_out/r3s_smi.txt
python3 -m pytest tests/test_gpu_long_sequences.py -x -q 2>&1 | tail -3
# sequences between a read and a contig: where should tiling start?
for L in 400 600 1000; do
  NC=$((30000000 / L))
  echo "== $L bp x $NC, tiles from 1,024 positions (default: none of these are tiled)"; python3 scripts/time_contigs.py $L $NC 2>&1 | tail -3
  echo "== $L bp x $NC, tiles from 256 positions"; KR_TILE_MIN_POS=256 python3 scripts/time_contigs.py $L $NC 2>&1 | tail -3
done
