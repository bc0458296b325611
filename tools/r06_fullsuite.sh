#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r06suite
mkdir -p $out
cd $GRAFT_REPO_ROOT
python3 bench.py --steps 20 --warmup 5 --no-e2e --no-cpu-baseline --no-dense > $out/bench_line.json 2> $out/bench_stderr.txt
python3 -c "
import json; d=json.load(open('$out/bench_line.json')); print(d['ms_per_step'], d['roofline']['profile_seed_launch']['avg_launch_ms'], d['roofline']['leaf_seed_launch']['avg_launch_ms'], d['hits_crc'])"
timeout 3300 python3 -m pytest tests -m gpu -x -q --durations=25 > $out/pytest_all.txt 2>&1
tail -45 $out/pytest_all.txt
