#!/bin/bash
# round 6, on the GPU box: per-kernel statistics of config C5's complete pipeline in the one-thread order (where its 400 s go)
out=$GRAFT_REPO_ROOT/gpurun_out/r06c5prof
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c5 -o c5 -- python3 $GRAFT_REPO_ROOT/bench.py --child e2e_c5 > $out/c5_leg.json 2> $out/stderr.txt
cp /tmp/prof_c5/c5_kernel_stats.csv $out/c5_kernel_stats.csv 2>/dev/null
ls -la $out; head -c 600 $out/c5_leg.json
