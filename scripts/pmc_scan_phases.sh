#!/bin/bash
# Wave-instruction counts of the scan kernel by phase: KR_DEBUG_SKIP ablations (4: front end and probe lists only, no table scan;
# 1: scan and hit test but no hit is resolved / emitted; 0: everything) under rocprofv3 --pmc SQ_INSTS_*; differences = phases.
export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8
for v in ${SKIPS:-0 1 4}; do
  rm -rf gpurun_out/pmcs_$v; mkdir -p gpurun_out/pmcs_$v
  KR_DEBUG_SKIP=$v rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d gpurun_out/pmcs_$v -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-host-inclusive --check-reads 1000 --reads-per-step 1000000 --read-procs 1 --distinct-batches 1 "$@" > gpurun_out/pmcs_$v/log.txt 2>&1
  python3 - <<PY
import csv,glob,collections
f=glob.glob("gpurun_out/pmcs_$v/**/*counter_collection.csv",recursive=True)[0]
m=collections.defaultdict(float)
t=collections.defaultdict(float)
for r in csv.DictReader(open(f)):
    if "kr_scan_" in r["Kernel_Name"]: m[r["Counter_Name"]]=max(m[r["Counter_Name"]], float(r["Counter_Value"]))
print("KR_DEBUG_SKIP=$v scan kernel, wave instructions per read:", {k: round(x/1e6,1) for k,x in sorted(m.items())})
PY
  find gpurun_out/pmcs_$v -name "*.csv" -delete
done
