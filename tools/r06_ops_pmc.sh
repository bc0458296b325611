#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r06ops
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/prof_ops -o ops -- python3 $GRAFT_REPO_ROOT/tools/bench_ops.py aa > $out/ops_table_aa.txt 2> $out/err.txt
cp /tmp/prof_ops/ops_counter_collection.csv $out/ops_aa_pmc_valu.csv
timeout 900 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS --output-format csv -d /tmp/prof_ops2 -o ops -- python3 $GRAFT_REPO_ROOT/tools/bench_ops.py aa > /dev/null 2>&1
cp /tmp/prof_ops2/ops_counter_collection.csv $out/ops_aa_pmc_mix.csv
cat $out/ops_table_aa.txt
