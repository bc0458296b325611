#!/bin/bash
# Runs on the GPU box: the wall-clock measurements of a round (NJ phase at several sizes, the join engine's kernel statistics at
# a million sequences, whole-pipeline comparisons with the compiled reference at config C2's size and on proteins, the tick
# counters of the quartet kernel when a timing build is present).  Output: gpurun_out/$1/.
tag=${1:-r03w}
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
# the tick counters need a variant build (VFT_EXTRA_HIPCC_FLAGS=-DVFT_ML_TIMING python -m veryfasttree_amd.build): loaded by path,
# the installed library is never touched
for v in build/variants/*/; do
  if grep -q VFT_ML_TIMING $v/FLAGS 2>/dev/null; then
    VFT_LIB_DIR=$GRAFT_REPO_ROOT/$v timeout 600 python3 tools/ml_ticks.py 1500 300 > $out/ml_ticks_aa_1500.txt 2>&1
  fi
done
cd /tmp && export TMPDIR=/tmp
timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_e1m -o e1m -- python3 $GRAFT_REPO_ROOT/tools/nj_gpu_only.py 1000000 200 mu=0.02 gap=0.01 seed=4 > $out/nj_1M_under_rocprof.log 2>&1
cp /tmp/prof_e1m/*kernel_stats.csv $out/engine_1M_kernel_stats.csv 2>/dev/null
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_aa -o aa -- python3 $GRAFT_REPO_ROOT/tools/full_pipeline_aa_once.py 3000 300 > $out/aa_3000_under_rocprof.log 2>&1
cp /tmp/prof_aa/*kernel_stats.csv $out/aa_3000_kernel_stats.csv 2>/dev/null
ls -la $out
