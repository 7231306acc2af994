#!/bin/bash
# Build an experiment variant of the library: scripts/build_variant.sh <name> <extra hipcc flags...>
# -> krepp_amd/lib/variants/<name>/libkrepp_amd.so   (host objects are shared with the main build)
set -e
NAME=$1; shift
cd /root/repo/krepp_amd/csrc
mkdir -p build/var_$NAME ../lib/variants/$NAME
/opt/rocm/bin/hipcc -std=c++17 -O3 -Wno-unused-value -fPIC -fvisibility=hidden --offload-arch=gfx950 -ffp-contract=off -I../../include -I. "$@" -c kr_device.hip -o build/var_$NAME/kr_device.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/variants/$NAME/libkrepp_amd.so build/var_$NAME/kr_device.o build/kr_minimizer.o build/kr_host.o build/kr_build.o build/kr_place.o -lz -lgomp -ldl
echo built $NAME
