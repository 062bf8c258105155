#!/bin/bash
# experiment libraries (within-run A/B harnesses; never loaded by dove_amd)
set -euo pipefail
cd "$(dirname "$0")"
for f in *_exp.hip; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $f -o lib${f%.hip}.so &
done
wait
echo built
