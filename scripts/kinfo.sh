#!/bin/bash
# Registers / LDS / scratch of every kernel of kr_device.hip whose mangled name matches $1 (default: scan); extra hipcc flags after it.
PAT=${1:-scan}; shift
cd /root/repo/krepp_amd/csrc
mkdir -p /tmp/kinfo
/opt/rocm/bin/hipcc -std=c++17 -O3 -Wno-unused-value --offload-arch=gfx950 -ffp-contract=off -I../../include -I. "$@" -S --cuda-device-only kr_device.hip -o /tmp/kinfo/d.s 2>&1 | grep -v hip-link
python3 - "$PAT" <<'PY'
import re, sys
s = open('/tmp/kinfo/d.s').read()
for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', s, re.S):
    name, body = m.group(1), m.group(2)
    if sys.argv[1] not in name: continue
    g = lambda k: (re.search(r'\.amdhsa_' + k + r'\s+(\S+)', body) or [None, '?'])[1]
    short = re.sub(r'_ZN12_GLOBAL__N_1\d+', '', name)[:64]
    print(f"{short:66s} vgpr={g('next_free_vgpr'):>4} sgpr={g('next_free_sgpr'):>4} lds={g('group_segment_fixed_size'):>6} scratch={g('private_segment_fixed_size'):>5}")
PY
