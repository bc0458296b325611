#!/bin/bash
# round 6, on the GPU box: instruction-issue counters of the batch operators under LG (tools/bench_ops.py aa) - what section 5m's kernels do with the SIMDs
out=$GRAFT_REPO_ROOT/gpurun_out/r06opspmc
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/prof_ops_valu -o ops -- python3 $GRAFT_REPO_ROOT/tools/bench_ops.py aa > $out/ops_table_aa.txt 2> /dev/null
cp /tmp/prof_ops_valu/ops_counter_collection.csv $out/ops_aa_pmc_valu.csv 2>/dev/null
ls -la $out
