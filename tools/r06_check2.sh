#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r06k
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_gpu_tile_streams.py tests/test_gpu_parity.py -m gpu -x -q > $out/pytest1.txt 2>&1
tail -3 $out/pytest1.txt
timeout 600 python3 -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "seeds_sharing or sweep_properties" > $out/pytest2.txt 2>&1
tail -3 $out/pytest2.txt
python3 bench.py --steps 20 --warmup 5 --no-e2e > $out/bench_line.json 2> $out/bench_stderr.txt
cut -c1-300 $out/bench_line.json
python3 tools/r06_thr200k.py $out/thr200k.tree > $out/thr200k.txt 2>&1; tail -2 $out/thr200k.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-e2e --no-cpu-baseline --no-dense > $out/bench_line_under_rocprof.json 2> $out/stats_stderr.txt
cp /tmp/prof_stats/bench_kernel_stats.csv $out/bench_kernel_stats.csv
