#!/bin/bash
# Dump the ISA of one kernel instantiation to /tmp/kk/k.s (default: the scan kernel, LOG_G=2, CPL=4, SL, no tap, slotted)
mkdir -p /tmp/kk
cd /root/repo/krepp_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I../../include -Wno-unused-value -S --cuda-device-only kr_device.hip -o /tmp/kk/d.s 2>&1 | grep -v hip-link
K=${1:-_ZN12_GLOBAL__N_116kr_scan_kernel_tILi2ELi4ELb1ELb0ELb1EEEvNS_8DevIndexENS_9DevParamsENS_7BatchInENS_8BatchOutE}
awk "/^$K:/,/s_endpgm/" /tmp/kk/d.s > /tmp/kk/k.s
wc -l /tmp/kk/k.s; echo "scratch ops: $(grep -c scratch_ /tmp/kk/k.s)"
grep -E "^\s+\.(sgpr|vgpr)_(count|spill_count)|scratch_en|private_segment_fixed_size" /tmp/kk/d.s | head -0
