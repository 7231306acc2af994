#!/bin/bash
# Round 3, session U: addresses of a stream's buffers next to its scan level; does every move of the item list / the counters draw a new level?
ulimit -c 0
cd "$(dirname "$0")/.."
export GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out
for rep in 1 2 3; do
  python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-host-inclusive --stream-variance 6 --stream-variance-move 2,0,2,0 \
     > gpurun_out/r3u_move$rep.json 2> gpurun_out/r3u_move$rep.err
  grep "stream-variance" gpurun_out/r3u_move$rep.err
  tail -2 gpurun_out/r3u_move$rep.err | cut -c1-200
done
