#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r06n
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests/test_gpu_ml_bits.py tests/test_gpu_parity.py tests/test_gpu_aa.py tests/test_gpu_chains.py -m gpu -x -q > $out/pytest_ml.txt 2>&1
tail -5 $out/pytest_ml.txt
python3 tools/bench_ops.py aa > $out/ops_table_aa.txt 2>&1; cat $out/ops_table_aa.txt
