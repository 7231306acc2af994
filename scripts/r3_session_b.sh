#!/bin/bash
# round 3, session B: the pipelined scan -- parity first, then launch times of the variants at several occupancies
mkdir -p gpurun_out
python -m pytest tests/test_gpu_place_k27.py tests/test_gpu_rccl_cli.py "tests/test_gpu_syn1000.py::test_syn1000_10gb_index_vs_oracle_and_full_batch_properties[slotted_w64]" tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r3b_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r3b_tests.log
tail -8 gpurun_out/r3b_tests.log
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-host-inclusive --check-reads 2000 --distinct-batches 1"
one() { # name, env...
  name=$1; shift
  echo -n "$name: "
  env "$@" $B 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), {k:round(v,2) for k,v in d['kernel_ms'].items() if k in ('scan','accumulate','llh_select')}, d['check']['rows_equal'])"
}
cp krepp_amd/lib/libkrepp_amd.so /tmp/main_lib.so
one old_w4 KR_SCAN_PIPE=0
one old_b3 KR_SCAN_PIPE=0 KR_DEBUG_SCAN_BLOCKS_PER_CU=3
one old_b2 KR_SCAN_PIPE=0 KR_DEBUG_SCAN_BLOCKS_PER_CU=2
for v in d1w4 d2w4 d2w3 d3w3 d4w2; do
  cp krepp_amd/lib/variants/$v/libkrepp_amd.so krepp_amd/lib/libkrepp_amd.so
  one $v X=1
  one ${v}_b3 KR_DEBUG_SCAN_BLOCKS_PER_CU=3
  one ${v}_b2 KR_DEBUG_SCAN_BLOCKS_PER_CU=2
done
cp /tmp/main_lib.so krepp_amd/lib/libkrepp_amd.so
