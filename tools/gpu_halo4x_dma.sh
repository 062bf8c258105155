#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do for v in 0 1; do DOVE_HALO4X_DMA=$v timeout 200 python tools/halo4x_dma.py 2>&1 | grep "DOVE_HALO4X" ; done; done | tee gpurun_out/r03_halo4x_dma.log
