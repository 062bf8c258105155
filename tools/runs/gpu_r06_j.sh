#!/bin/bash
# round 6, call J: where the tiled VAE loses against its FLOP ratio - rocprofv3 kernel stats of the tiled and the untiled VAE, one stream each
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for m in untiled tiled; do
  cd /tmp && rm -rf /tmp/prof_$m && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$m -o vae -- python $R/tools/vae_mode_profile.py --mode $m > $R/gpurun_out/r06_j_$m.log 2>&1
  cp /tmp/prof_$m/vae_kernel_stats.csv $R/gpurun_out/r06_j_${m}_kernel_stats.csv
done
cd $R
tail -25 gpurun_out/r06_j_tiled.log
