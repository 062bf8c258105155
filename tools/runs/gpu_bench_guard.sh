#!/bin/bash
# bench.py's guard around the single-clip mode, exercised with 2 ranks on GPU 0 (gloo): a clean run, a rank that raises, a rank that hangs.
export TMPDIR=/tmp
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
L=gpurun_out/r03_bench_guard.log
: > $L
run() { echo "== $1" >> $L; shift; ( "$@" timeout -k 10 600 python bench.py --gpus 2 --oversubscribe --layers 2 --steps 1 --warmup 1 2>&1 | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); sc = d.get('single_clip', {})
    print('value', round(d['value'], 2), 'n_gpus', d['n_gpus'], '| single_clip:', {k: sc[k] for k in ('value', 'error', 'frames_decoded_per_rank', 'efficiency_vs_n1') if k in sc})
" ) >> $L 2>&1; echo "exit $?" >> $L; }
run "clean" env
run "rank 1 raises" env DOVE_BENCH_STRONG_FAULT=raise:1 DOVE_BENCH_STRONG_TIMEOUT=60
run "rank 0 raises" env DOVE_BENCH_STRONG_FAULT=raise:0 DOVE_BENCH_STRONG_TIMEOUT=60
run "rank 1 hangs" env DOVE_BENCH_STRONG_FAULT=hang:1 DOVE_BENCH_STRONG_TIMEOUT=45
cat $L
