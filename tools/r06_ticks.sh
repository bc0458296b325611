#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r06ticks
mkdir -p $out
cd $GRAFT_REPO_ROOT
VFT_LIB_DIR=$GRAFT_REPO_ROOT/build/njtiming python3 tools/engine_ticks.py 1000000 200 > $out/engine_ticks_1M.txt 2>&1
cat $out/engine_ticks_1M.txt
