#!/bin/bash
# On the GPU box: bench every variant under krepp_amd/lib/variants (the main library is restored last).
cp krepp_amd/lib/libkrepp_amd.so /tmp/main_lib.so
for d in krepp_amd/lib/variants/*/; do
  v=$(basename $d)
  cp $d/libkrepp_amd.so krepp_amd/lib/libkrepp_amd.so
  echo -n "$v: "
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --check-reads 2000 "$@" 2>&1 | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2), d['kernel_ms'], d['check']['rows_equal'])"
done
cp /tmp/main_lib.so krepp_amd/lib/libkrepp_amd.so
