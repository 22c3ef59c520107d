#!/bin/bash
# Register / scratch / LDS summary of the fullrank_verify_kernel instantiations of one part: tools/isa_fv.sh <part> [extra flags]
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
P=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math --cuda-device-only -S -DFV_PART=$P "$@" \
    "$ROOT/fastrank_amd/csrc/fullverify.hip" -o /tmp/fv_$P.s 2>&1 | grep -v "warning\|^$" || true
python3 - /tmp/fv_$P.s <<'PY'
import re, sys, subprocess
txt = open(sys.argv[1]).read()
for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', txt, re.S):
    name, body = m.group(1), m.group(2)
    g = lambda k: (re.search(r'\.amdhsa_' + k + r' (\S+)', body) or [None, None])[1]
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r"\(.*", "", dem.replace("void frdev::", ""))
    # instruction count of the kernel body
    b = re.search(re.escape(name) + r":\n(.*?)s_endpgm", txt, re.S)
    ninst = len([l for l in b.group(1).splitlines() if l.startswith("\t") and not l.strip().startswith((".", ";"))]) if b else -1
    print("%-50s vgpr %s sgpr %s scratch %s insts %d" % (dem, g('next_free_vgpr'), g('next_free_sgpr'), g('private_segment_fixed_size'), ninst))
PY
