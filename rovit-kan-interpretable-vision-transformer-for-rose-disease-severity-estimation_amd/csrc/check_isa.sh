#!/bin/bash
# Build-time check on the emitted gfx950 ISA of librovit_hip.so (called by the Makefile after linking).
#
# Rule: no packed-fp32 VALU instruction (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32) anywhere in the device code.
# Why (DESIGN.md "packed-fp32 hazard"): hipcc's SLP vectoriser turns adjacent scalar fp32 adds into v_pk_add_f32 with
# op_sel half-swaps; one such instruction in the residual+LayerNorm GEMM epilogue lost the low half of its result in
# lanes 48-63 in ~15 % of launches at M = 50432 whenever a second workgroup on the CU was issuing MFMAs
# (tools/hazard/: in-kernel A/B builds and a stand-alone register-only replay).  The library is built with
# -fno-slp-vectorize; this script makes sure a compiler or flag change cannot silently bring the instructions back.
set -e
LIB="$1"
OBJDUMP=${OBJDUMP:-/opt/rocm/lib/llvm/bin/llvm-objdump}
TMP=$(mktemp -d)
trap 'rm -rf "$TMP"' EXIT
cp "$LIB" "$TMP/lib.so"
(cd "$TMP" && "$OBJDUMP" --offloading lib.so > /dev/null)
n_code=0; n_pk=0
for f in "$TMP"/lib.so.*gfx950*; do
  [ -f "$f" ] || continue
  "$OBJDUMP" -d "$f" > "$TMP/dis.txt"
  n_code=$((n_code + $(grep -c 's_endpgm' "$TMP/dis.txt" || true)))
  n_pk=$((n_pk + $(grep -cE 'v_pk_(add|mul|fma)_f32' "$TMP/dis.txt" || true)))
done
if [ "$n_code" -eq 0 ]; then echo "check_isa: no gfx950 code objects found in $LIB" >&2; exit 1; fi
if [ "$n_pk" -ne 0 ]; then
  echo "check_isa: $n_pk packed-fp32 VALU instructions in $LIB -- build with -fno-slp-vectorize (see DESIGN.md)" >&2; exit 1
fi
echo "check_isa: ok ($n_code kernels, no packed-fp32 VALU instructions)"
