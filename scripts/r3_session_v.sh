#!/bin/bash
# Round 3, session V: what the waves of a slow scan launch wait for -- counter passes over ONE process each with fast and slow streams in it
ulimit -c 0
cd "$(dirname "$0")/.."
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=8
mkdir -p gpurun_out
pass() {
  name=$1; shift
  OUT=$PWD/gpurun_out/r3v_$name
  rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-host-inclusive --read-procs 1 --check-reads 2000 --distinct-batches 1 --stream-variance 7 > $OUT/bench.log 2>&1
  echo "== $name: $*"
  python3 - $OUT <<'PY'
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)
if not f:
    print('no counter file'); print(open(sys.argv[1] + '/bench.log').read()[-1500:]); sys.exit(0)
rows = collections.OrderedDict()
for r in csv.DictReader(open(f[0])):
    if 'kr_scan' not in r['Kernel_Name']: continue
    k = int(r['Dispatch_Id'])
    d = rows.setdefault(k, {'dur': (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6})
    d[r['Counter_Name']] = d.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
for k, v in rows.items():
    if v['dur'] < 5: continue
    print(k, {a: (round(b, 2) if a == 'dur' else f'{b:.5g}') for a, b in v.items()})
PY
  rm -rf /tmp/krepp_bench_* 
}
pass icache SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL
pass wait SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES
pass insts SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_LDS
pass wr TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum
pass atomic TCC_EA0_ATOMIC_sum TCC_EA0_ATOMIC_LEVEL_sum TCC_ATOMIC_sum SQC_DCACHE_MISSES SQC_DCACHE_REQ
pass cyc SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS
