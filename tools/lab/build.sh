#!/bin/bash
# builds the lab programs of tools/lab/ against the in-tree library (run from the repo root; binaries in tools/lab/bin/, git-ignored)
P=$(ls -d rovit-kan*/)
mkdir -p tools/lab/bin
for src in tools/lab/*.hip; do
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-slp-vectorize -fno-vectorize -Wno-unused-value -Wno-unused-result -I ${P}csrc -I include $src \
        -L${P}lib -lrovit_hip -Wl,-rpath,'$ORIGIN/../../../'${P}lib -o tools/lab/bin/$(basename $src .hip) || exit 1
done
