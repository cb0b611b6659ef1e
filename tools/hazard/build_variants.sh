#!/bin/bash
# Builds A/B variants of librovit_hip.so that re-create the round-1 epilogue of the fused residual+LayerNorm GEMM
# (see DESIGN.md "observed hazard") under tools/hazard/build/ from the frozen round-2 copy tools/hazard/gemm_r02_repro.hip
# (the product's csrc/gemm.hip no longer carries the repro blocks); the other objects are the product build's (run
# `make -C csrc` first).
set -e
HERE=$(cd "$(dirname "$0")" && pwd)
CSRC="$HERE/../../rovit-kan-interpretable-vision-transformer-for-rose-disease-severity-estimation_amd/csrc"
OUT="$HERE/build"; mkdir -p "$OUT"
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function"
OTHERS=$(ls "$CSRC"/*.o | grep -v gemm.o)
build() {   # name, extra flags
  /opt/rocm/bin/hipcc $FLAGS $2 -I "$CSRC" -c "$HERE/gemm_r02_repro.hip" -o "$OUT/gemm_$1.o"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OTHERS "$OUT/gemm_$1.o" -o "$OUT/librovit_$1.so"
  rm -f "$OUT/gemm_$1.o"
}
build v1 "-DROVIT_HAZARD_REPRO=1" &
build v2 "-DROVIT_HAZARD_REPRO=2" &
build v3 "-DROVIT_HAZARD_REPRO=3" &
build v4 "-DROVIT_HAZARD_REPRO=1 -fno-slp-vectorize" &
build v5 "-DROVIT_HAZARD_REPRO=5" &
build v6 "-DROVIT_HAZARD_REPRO=6" &
wait
ls -la "$OUT"
