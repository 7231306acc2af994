#!/bin/bash
# AddressSanitizer over the host C++ (reader, builder, placement) with the CPU test suite; the device objects are linked as built.
# (GPU sanitizers are not available on the pool; this covers what runs on the CPU.)  Restores the normal library afterwards.
set -e
cd "$(dirname "$0")/../krepp_amd/csrc"
mkdir -p /tmp/asan
for f in kr_host kr_build kr_place; do
  g++ -std=c++17 -O1 -g -fPIC -fvisibility=hidden -fopenmp -fsanitize=address -fno-omit-frame-pointer -I../../include -I. -c $f.cpp -o /tmp/asan/$f.o
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/asan/libkrepp_amd.so build/kr_device.o build/kr_minimizer.o /tmp/asan/kr_host.o /tmp/asan/kr_build.o /tmp/asan/kr_place.o -lz -lgomp -ldl
cd ../..
cp krepp_amd/lib/libkrepp_amd.so /tmp/asan/orig.so
trap 'cp /tmp/asan/orig.so krepp_amd/lib/libkrepp_amd.so' EXIT # whatever happens below, the product library comes back
cp /tmp/asan/libkrepp_amd.so krepp_amd/lib/libkrepp_amd.so
# the suite's exit code is this script's; the last lines of its output go to profiles/ (usage: scripts/asan_host.sh [summary file])
SUMMARY=${1:-profiles/asan_host_latest.txt}
set +e
LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0:abort_on_error=0 python -m pytest tests -m "not gpu" -x -q -p no:cacheprovider > /tmp/asan/pytest.log 2>&1
RC=$?
set -e
{
  echo "scripts/asan_host.sh: host C++ (kr_host.cpp, kr_build.cpp, kr_place.cpp) under -fsanitize=address, CPU test suite (pytest -m 'not gpu')"
  echo "commit $(git rev-parse --short HEAD)$(git diff --quiet || echo ' + uncommitted changes'), $(date -u +%Y-%m-%dT%H:%MZ), exit code $RC"
  echo "AddressSanitizer reports in the log: $(grep -c 'ERROR: AddressSanitizer' /tmp/asan/pytest.log || true)"
  tail -5 /tmp/asan/pytest.log
} > "$SUMMARY"
cat "$SUMMARY"
exit $RC
