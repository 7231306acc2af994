#!/bin/bash
# Dump the ISA of one probe-kernel instantiation to /tmp/kk/k.s (default: LOG_G=2, CPL=3, SL, no tap)
mkdir -p /tmp/kk
cd /root/repo/krepp_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I../../include -Wno-unused-value -S --cuda-device-only kr_device.hip -o /tmp/kk/d.s 2>&1 | grep -v hip-link
K=${1:-_ZN12_GLOBAL__N_117kr_probe_kernel_tILi2ELi3ELb1ELb0EEEvNS_8DevIndexENS_9DevParamsENS_7BatchInENS_8BatchOutE}
awk "/^$K:/,/s_endpgm/" /tmp/kk/d.s > /tmp/kk/k.s
wc -l /tmp/kk/k.s; echo "scratch ops: $(grep -c scratch_ /tmp/kk/k.s)"
grep -E "^\s+\.(sgpr|vgpr)_(count|spill_count)|scratch_en|private_segment_fixed_size" /tmp/kk/d.s | head -0
