#!/bin/bash
# round 6: the profile set at HEAD (digests current), bench lines (default flags, the driver's flags), the 10,000-genome index, place CLI.
# usage (on the GPU box, from the repo root): bash scripts/r6_final_session.sh <tag>
TAG=${1:-r6a}
ulimit -c 0
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash scripts/profile.sh $TAG > gpurun_out/${TAG}_profile.log 2>&1
python3 bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python3 scripts/traffic.py gpurun_out/prof_$TAG gpurun_out/${TAG}_bench.json gpurun_out/${TAG}_traffic.json > gpurun_out/${TAG}_traffic.log 2>&1
cp gpurun_out/${TAG}_traffic.json profiles/traffic_latest.json   # (on the box: the driver-flag run below reports it; copied into the repo from gpurun_out afterwards)
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driver.json 2> gpurun_out/${TAG}_bench_driver.err
tail -c 400 gpurun_out/${TAG}_bench_driver.json; echo
bash scripts/profile.sh ${TAG}_s10k --workload syn10000 > gpurun_out/${TAG}_s10k_profile.log 2>&1
python3 bench.py --workload syn10000 --no-cpu-baseline > gpurun_out/${TAG}_s10k_bench.json 2> gpurun_out/${TAG}_s10k_bench.err
tail -c 300 gpurun_out/${TAG}_s10k_bench.json; echo
if [ -z "$KR_FINAL_NO_PLACE" ]; then
  timeout 600 python scripts/time_cli_place_big.py 4000000 > gpurun_out/${TAG}_cli_place_big.txt 2>&1
  tail -12 gpurun_out/${TAG}_cli_place_big.txt
fi
# what comes back is at most 64 MiB: the summaries and the traced runs' bench lines, not the profiler's tables
for t in $TAG ${TAG}_s10k; do
  cp gpurun_out/prof_$t/summary_$t.txt gpurun_out/${t}_rocprof_summary.txt
  cp gpurun_out/prof_$t/summary_$t.json gpurun_out/${t}_rocprof_summary.json
  grep '^{"metric"' gpurun_out/prof_$t/bench_trace.log | tail -1 > gpurun_out/${t}_bench_line_traced_run.json
  f=$(grep -l kr_scan_pipe $(find gpurun_out/prof_$t/trace -name "*kernel_stats.csv") | head -n 1) # (the bench's own process, not its helpers')
  [ -n "$f" ] && cp "$f" gpurun_out/${t}_kernel_stats.csv
  rm -rf gpurun_out/prof_$t
done
du -sh gpurun_out
