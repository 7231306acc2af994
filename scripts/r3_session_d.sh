#!/bin/bash
# round 3, session D: place tests (heavy reads on the device), counter list, stream-to-stream variance of the pipelined scan
mkdir -p gpurun_out
python -m pytest tests/test_place.py tests/test_gpu_place_k27.py tests/test_gpu_rccl_cli.py tests/test_gpu_syn1000.py::test_place_on_the_1000_genome_tree_never_leaves_the_device -m gpu -x -q -s > gpurun_out/r3d_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r3d_tests.log
tail -8 gpurun_out/r3d_tests.log; grep "heavy reads" gpurun_out/r3d_tests.log
cd /tmp && export TMPDIR=/tmp && rocprofv3 -L > $GRAFT_REPO_ROOT/gpurun_out/r3d_counters.txt 2>&1; cd $GRAFT_REPO_ROOT
grep -c . gpurun_out/r3d_counters.txt; grep -i "TCC_EA0_RDREQ\b\|TCC_EA0_WRREQ\b\|TCC_REQ\b\|TCP_TCC_READ_REQ\b\|MALL\|TCC_EA0_RD_UNCACHED" gpurun_out/r3d_counters.txt | head -20
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-host-inclusive --check-reads 2000 --distinct-batches 1 --stream-variance 8"
for i in 1 2; do $B 2> gpurun_out/r3d_var$i.err | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), {k:round(v,2) for k,v in d['kernel_ms'].items() if k in ('scan','accumulate','llh_select')})"; grep stream-variance gpurun_out/r3d_var$i.err | sed 's/\[stream-variance\] //'; done
