#!/bin/bash
# Wave-instruction counts of the lean accumulate kernel by phase: KR_DEBUG_SKIP ablations (2: no items, 16: no events kept,
# 64: no keys, 128: no plane pass, 256: no record output) under rocprofv3 --pmc SQ_INSTS_*; differences between rows = phases.
export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8 # before the profiler's preload initialises the runtime
for v in ${SKIPS:-0 256 128 64 16 2}; do
  rm -rf gpurun_out/pmci_$v; mkdir -p gpurun_out/pmci_$v
  KR_DEBUG_SKIP=$v rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d gpurun_out/pmci_$v -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-host-inclusive --check-reads 1000 --reads-per-step 1000000 --read-procs 1 --distinct-batches 1 > gpurun_out/pmci_$v/log.txt 2>&1
  python3 - <<PY
import csv,glob,collections
m=collections.defaultdict(float)
for f in glob.glob("gpurun_out/pmci_$v/**/*counter_collection.csv",recursive=True):  # (one file per process: the index inflation runs in a child)
    for r in csv.DictReader(open(f)):
        if "kr_acc_kernel_t<true, 5, false, 7" in r["Kernel_Name"]: m[r["Counter_Name"]]=max(m[r["Counter_Name"]], float(r["Counter_Value"]))
print("KR_DEBUG_SKIP=$v per read:", {k: round(x/1e6,1) for k,x in sorted(m.items())})
PY
  find gpurun_out/pmci_$v -name "*.csv" -delete
done
