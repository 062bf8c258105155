#!/bin/bash
# Build libdove_hip.so in-tree for gfx950 (hipcc cross-compiles without a GPU).
#   build.sh            -> ../libdove_hip.so          (the product library)
#   build.sh timing     -> ../libdove_hip_timing.so   (-DDOVE_TIMING_BUILD: ablation switches + s_memtime phase logs for
#                                                      tools/*_timing.py and tools/microbench.py; never loaded by dove_amd)
set -euo pipefail
cd "$(dirname "$0")"
MODE="${1:-product}"
OUT=../libdove_hip.so
OBJ=.
EXTRA=""
if [ "$MODE" = "timing" ]; then OUT=../libdove_hip_timing.so; OBJ=.timing; EXTRA="-DDOVE_TIMING_BUILD"; mkdir -p $OBJ; fi
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result $EXTRA"
SRCS="capi igemm igemm_legacy norm attention attention_pipe attention_mx elementwise mxfp8 t5 graph"
if [ "$MODE" = "timing" ]; then SRCS="$SRCS gemm4x_timing"; fi
pids=()
for f in $SRCS; do
  hipcc $FLAGS -c $f.hip -o $OBJ/$f.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
objs=""
for f in $SRCS; do objs="$objs $OBJ/$f.o"; done
hipcc --offload-arch=gfx950 -shared -fPIC $objs -o $OUT
echo "built $(realpath $OUT)"
