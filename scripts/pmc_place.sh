#!/bin/bash
# Instruction counters of the place kernels on the 1000-genome tree (scripts/time_place_big.py): usage scripts/pmc_place.sh <tag>
TAG=${1:-pl}
export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8
OUT=gpurun_out/pmcpl_$TAG
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $OUT -- python3 scripts/time_place_big.py > $OUT/log.txt 2>&1
python3 - <<PY
import csv,glob,collections,re
d=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        m=re.search(r'(kr_place\w*)(<[^>]*>)?', r["Kernel_Name"])
        if m: d[m.group(1)+(m.group(2) or "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in d.items():
    print(k, {c:(len(x), round(max(x))) for c,x in v.items()})
PY
tail -5 $OUT/log.txt
find $OUT -name "*.csv" -size +5M -delete
