#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r06m
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests/test_gpu_multirank.py -m gpu -x -q > $out/pytest_multi.txt 2>&1
tail -5 $out/pytest_multi.txt
timeout 1500 python3 -m pytest tests/test_gpu_walk_server.py tests/test_gpu_tophits.py tests/test_gpu_nj_driver.py tests/test_gpu_stress.py -m gpu -x -q > $out/pytest_b.txt 2>&1
tail -5 $out/pytest_b.txt
