#!/bin/bash
# round 3, session F: what the slow scan launches are (address-translation counters), item chunk size, likelihood stage beside the next scan
mkdir -p gpurun_out
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-host-inclusive --check-reads 2000 --distinct-batches 1"
one() { name=$1; shift; echo -n "$name: "; env "$@" 2>gpurun_out/r3f_$name.err | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), round(d['ms_per_step'],2), {k:round(v,2) for k,v in d['kernel_ms'].items() if k in ('scan','accumulate','llh_select')}, d['check']['rows_equal'])"; grep stream-variance gpurun_out/r3f_$name.err | sed 's/\[stream-variance\] //'; }
# --- overlap mode 2: dedup .. select of batch i beside the scan of batch i+1
one serial X=1 $B
one ovl2_d1_s3 KR_OVERLAP=2 KR_OVERLAP_SCAN_D1=1 KR_OVERLAP_SCAN_BLOCKS=3 $B --pipeline-streams 2
one ovl2_d1_s4 KR_OVERLAP=2 KR_OVERLAP_SCAN_D1=1 KR_OVERLAP_SCAN_BLOCKS=4 $B --pipeline-streams 2
one ovl2_d2_s2 KR_OVERLAP=2 KR_OVERLAP_SCAN_BLOCKS=2 $B --pipeline-streams 2
one ovl2_d2_s3 KR_OVERLAP=2 KR_OVERLAP_SCAN_BLOCKS=3 $B --pipeline-streams 2
# --- item chunk size against the stream-to-stream levels of the scan
cp krepp_amd/lib/libkrepp_amd.so /tmp/main_lib.so
for v in chunk512 chunk256; do
  cp krepp_amd/lib/variants/$v/libkrepp_amd.so krepp_amd/lib/libkrepp_amd.so
  one ${v}_a X=1 $B --stream-variance 4
  one ${v}_b X=1 $B --stream-variance 4
done
cp /tmp/main_lib.so krepp_amd/lib/libkrepp_amd.so
one chunk2048_a X=1 $B --stream-variance 4
# --- address translation counters per scan launch (fast and slow streams in one process)
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8
OUT=$PWD/gpurun_out/r3f_tlb
mkdir -p $OUT
rocprofv3 --pmc TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_THRASHING_STALL_sum --kernel-trace --output-format csv -d $OUT -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-inclusive --read-procs 1 --check-reads 2000 --distinct-batches 1 --stream-variance 5 > $OUT/bench.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/r3f_tlb/**/*counter_collection.csv', recursive=True)
print(f)
rows = collections.OrderedDict()
for r in csv.DictReader(open(f[0])):
    if 'kr_scan' not in r['Kernel_Name']: continue
    k = int(r['Dispatch_Id'])
    rows.setdefault(k, {'dur': (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6})[r['Counter_Name']] = float(r['Counter_Value'])
for k, v in rows.items():
    print(k, {a: (round(b, 2) if a == 'dur' else f'{b:.4g}') for a, b in v.items()})
PY
