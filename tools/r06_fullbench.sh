#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r06bench
mkdir -p $out
cd $GRAFT_REPO_ROOT
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_stdout.txt 2> $out/bench_stderr.txt
echo "rc=$?"; tail -1 $out/bench_stdout.txt | cut -c1-300; wc -l $out/bench_stdout.txt; tail -5 $out/bench_stderr.txt
