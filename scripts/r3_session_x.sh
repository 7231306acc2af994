#!/bin/bash
# Round 3, session X: address-translation counters of the scan with default and with physically contiguous stream buffers
ulimit -c 0
cd "$(dirname "$0")/.."
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out
pass() {
  name=$1; mode=$2; shift; shift
  OUT=$PWD/gpurun_out/r3x_$name
  rm -rf $OUT; mkdir -p $OUT
  export KR_HBM_CONTIGUOUS=$mode
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-inclusive --read-procs 1 --check-reads 2000 --distinct-batches 1 --stream-variance 6 > $OUT/bench.log 2>&1
  echo "== $name (KR_HBM_CONTIGUOUS=$mode): $*"
  python3 - $OUT <<'PY'
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)
if not f:
    print('no counter file'); print(open(sys.argv[1] + '/bench.log').read()[-1500:]); sys.exit(0)
rows = collections.OrderedDict()
for r in csv.DictReader(open(f[0])):
    kn = r['Kernel_Name']
    tag = 'scan' if 'kr_scan' in kn else ('acc' if 'kr_acc_kernel_t<true, 5, false, 7>' in kn else ('select' if 'kr_select' in kn else None))
    if not tag: continue
    k = (int(r['Dispatch_Id']), tag)
    d = rows.setdefault(k, {'dur': (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6})
    d[r['Counter_Name']] = d.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
for k, v in rows.items():
    if v['dur'] < 3: continue
    print(k[0], k[1], ' '.join(f"{a.replace('TCP_UTCL1_', '').replace('_sum', '')}={(round(b, 2) if a == 'dur' else format(b, '.4g'))}" for a, b in v.items()))
PY
  rm -rf /tmp/krepp_bench_*
}
C1="TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_THRASHING_STALL_sum"
C2="TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_SERIALIZATION_STALL_sum"
pass tlb_default 0 $C1
pass tlb_contig 2 $C1
pass stall_default 0 $C2
pass stall_contig 2 $C2
