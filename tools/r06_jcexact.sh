#!/bin/bash
# round 6, on the GPU box: the 100 000-sequence / 64-thread pipeline with the Jukes-Cantor totals as they were and bit for bit the reference's
out=$GRAFT_REPO_ROOT/gpurun_out/r06jc
mkdir -p $out
cd $GRAFT_REPO_ROOT
python3 tools/r06_thr200k.py $out/ours_100k_t64_fast.tree 1000 100000 > $out/fast.txt 2>&1; tail -2 $out/fast.txt
VFT_JC_EXACT=1 python3 tools/r06_thr200k.py $out/ours_100k_t64_exact.tree 1000 100000 > $out/exact.txt 2>&1; tail -2 $out/exact.txt
timeout 1200 python3 -m pytest tests/test_gpu_ml_long.py -m gpu -x -q --durations=10 > $out/pytest_long.txt 2>&1; tail -15 $out/pytest_long.txt
