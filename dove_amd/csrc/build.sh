#!/bin/bash
# Build libdove_hip.so in-tree for gfx950 (hipcc cross-compiles without a GPU).
set -euo pipefail
cd "$(dirname "$0")"
OUT=../libdove_hip.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result"
pids=()
for f in capi igemm norm attention elementwise; do
  hipcc $FLAGS -c $f.hip -o $f.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC capi.o igemm.o norm.o attention.o elementwise.o -o $OUT
echo "built $(realpath $OUT)"
