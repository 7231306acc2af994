export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8 # before the profiler's preload initialises the runtime
for v in ${SKIPS:-0 256 64 16 2}; do
  rm -rf gpurun_out/pmcw_$v; mkdir -p gpurun_out/pmcw_$v
  KR_DEBUG_SKIP=$v rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmcw_$v -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-host-inclusive --check-reads 1000 --reads-per-step 1000000 --read-procs 1 --distinct-batches 1 > gpurun_out/pmcw_$v/log.txt 2>&1
  python3 - <<PY
import csv,glob,collections
f=glob.glob("gpurun_out/pmcw_$v/**/*counter_collection.csv",recursive=True)[0]
m=collections.defaultdict(float)
for r in csv.DictReader(open(f)):
    n=r["Kernel_Name"]
    if "kr_acc_kernel_t<true, 5, false>" in n: m["acc_lean"]=max(m["acc_lean"], float(r["Counter_Value"]))
    if "kr_scan_kernel" in n: m["scan"]=max(m["scan"], float(r["Counter_Value"]))
print("KR_DEBUG_SKIP=$v WRITE_SIZE KB per 1M-read launch:", dict(m))
PY
  find gpurun_out/pmcw_$v -name "*.csv" -delete
done
