#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r06pins
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 2400 python3 -m pytest tests/test_gpu_fullsize.py tests/test_gpu_threads.py -m gpu -q -x --durations=15 -k "c4_join_order or c5_generator or 200_000" > $out/pytest_pins.txt 2>&1
tail -30 $out/pytest_pins.txt
