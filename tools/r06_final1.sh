#!/bin/bash
# round 6, on the GPU box: the pins (C4's complete trace, the 100 000-sequence / 64-thread fixture, treeLogLk split over two ranks), what one
# rank of W computes per step, the walk server's tick table, config C3's NJ phase by section, the join engine's kernel statistics at 1M
out=$GRAFT_REPO_ROOT/gpurun_out/r06final1
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests/test_gpu_fullsize.py tests/test_gpu_threads.py -m gpu -x -q --durations=10 \
  -k "c4_join_order or far_beyond or lanes_across" > $out/pytest_pins.txt 2>&1
tail -15 $out/pytest_pins.txt
timeout 600 python3 tools/shard_step_probe.py > $out/shard_step_probe.txt 2>&1
cat $out/shard_step_probe.txt
if [ -d build/walktiming ]; then
  for spec in "10000 200" "5000 1000" "5000 300 aa"; do
    VFT_LIB_DIR=$GRAFT_REPO_ROOT/build/walktiming timeout 600 python3 tools/walk_ticks.py $spec >> $out/walk_ticks.txt 2>&1
  done
fi
echo "---- production library" >> $out/walk_ticks.txt
for spec in "10000 200" "5000 1000" "5000 300 aa"; do
  timeout 600 python3 tools/walk_ticks.py $spec >> $out/walk_ticks.txt 2>&1
done
cat $out/walk_ticks.txt
VFT_NJ_PROFILE=1 timeout 600 python3 tools/nj_gpu_only.py 100000 500 fastest second mu=0.03 gap=0.01 seed=3 > $out/nj_c3_profile.txt 2>&1
tail -60 $out/nj_c3_profile.txt
cd /tmp && export TMPDIR=/tmp
timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_e1m -o e1m -- python3 $GRAFT_REPO_ROOT/tools/nj_gpu_only.py 1000000 200 mu=0.02 gap=0.01 seed=4 > $out/nj_1M_under_rocprof.log 2>&1
cp /tmp/prof_e1m/*kernel_stats.csv $out/engine_1M_kernel_stats.csv 2>/dev/null
tail -2 $out/nj_1M_under_rocprof.log
