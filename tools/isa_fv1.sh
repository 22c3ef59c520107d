#!/bin/bash
# ISA of ONE instantiation: tools/isa_fv1.sh <NL> <PL> <MODE> [extra flags]  -> /tmp/fv1.s + summary + where the scratch traffic is
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
NL=$1; PL=$2; MODE=$3; shift 3
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math --cuda-device-only -S -DFV_PART=0 -DFV_ONLY_NL=$NL -DFV_ONLY_PL=$PL -DFV_ONLY_MODE=$MODE "$@" \
    "$ROOT/fastrank_amd/csrc/fullverify.hip" -o /tmp/fv1.s 2>&1 | grep -v "warning\|^$" || true
grep -E "next_free_vgpr|private_segment_fixed_size|next_free_sgpr" /tmp/fv1.s | head -3
echo "scratch ops: $(grep -c 'scratch_\|buffer_store_dword.*offen\|buffer_load_dword.*offen' /tmp/fv1.s)  v_min_f64: $(grep -c v_min_f64 /tmp/fv1.s) bpermute: $(grep -c ds_bpermute /tmp/fv1.s) total lines: $(grep -c '^\s' /tmp/fv1.s)"
