#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/calib
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 scripts/calib_fetch.hip -o /tmp/calib_fetch || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/calib/fetch -- /tmp/calib_fetch > gpurun_out/calib/run.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/calib/trace -- /tmp/calib_fetch >> gpurun_out/calib/run.log 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/calib/fetch/**/*counter_collection.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "gather" in r["Kernel_Name"]: print(r["Kernel_Name"][:20], r["Counter_Name"], r["Counter_Value"], r.get("Dispatch_Id"))
f=glob.glob('gpurun_out/calib/trace/**/*kernel_trace.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'gather' in r['Kernel_Name']: print(r['Kernel_Name'][:20], 'ms', (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e6)
PY
grep known gpurun_out/calib/run.log | head -3
