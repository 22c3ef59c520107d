#!/bin/bash
# Device-only assembly of the HIP translation unit + register / scratch summary of kernels matching $1.
# usage: tools/isa.sh <kernel-name-substring> [extra hipcc flags]
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
PAT="$1"; shift || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math --cuda-device-only -S \
    "$ROOT/fastrank_amd/csrc/device.hip" -o /tmp/device.s "$@" 2>&1 | grep -v "warning\|^$" || true
python3 - "$PAT" <<'PY'
import re, sys
txt = open('/tmp/device.s').read()
for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', txt, re.S):
    name, body = m.group(1), m.group(2)
    if sys.argv[1] not in name:
        continue
    g = lambda k: (re.search(r'\.amdhsa_' + k + r' (\S+)', body) or [None, None])[1]
    print(name, 'vgpr', g('next_free_vgpr'), 'sgpr', g('next_free_sgpr'), 'scratch', g('private_segment_fixed_size'))
PY
