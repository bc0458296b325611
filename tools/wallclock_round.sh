#!/bin/bash
# Runs on the GPU box: the wall-clock measurements of a round (NJ phase at several sizes, whole-pipeline comparisons with the
# compiled reference at config C2's size and on proteins).  Output: gpurun_out/$1/.
tag=${1:-r02w}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
for spec in "20000 200" "20000 200 fastest" "100000 500 fastest" "300000 200"; do
  name=$(echo $spec | tr ' ' '_')
  timeout 900 python3 tools/nj_gpu_only.py $spec mu=0.02 gap=0.01 seed=4 > $out/nj_$name.log 2>&1
  tail -1 $out/nj_$name.log
done
VFT_NJ_PROFILE=1 timeout 1500 python3 tools/nj_gpu_only.py 1000000 200 mu=0.02 gap=0.01 seed=4 > $out/nj_1000000_200_profile.log 2>&1
tail -1 $out/nj_1000000_200_profile.log
timeout 900 python3 tests/compare_with_reference_run.py 10000 1000 -gtr --out $out/compare_c2.txt > /dev/null 2>&1
tail -8 $out/compare_c2.txt
timeout 900 python3 tests/compare_with_reference_run.py 3000 300 --aa -lg -double-precision --out $out/compare_aa_3000.txt > /dev/null 2>&1
tail -8 $out/compare_aa_3000.txt
