#!/bin/bash
# Runs on the GPU box (gpurun): the bench line, the rocprofv3 kernel statistics of the same command, the two PMC passes the
# HBM-traffic figure comes from (FETCH_SIZE and WRITE_SIZE in separate passes, MI355X_MICROARCH.md), and the per-operator
# tables.  Everything lands in gpurun_out/$1/; tools/profile_collect.py turns it into profiles/$1_* afterwards.
tag=${1:-r06}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
python3 bench.py --steps 10 --warmup 2 --no-e2e > $out/bench_line.json 2> $out/bench_stderr.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-e2e --no-cpu-baseline > $out/bench_line_under_rocprof.json 2> /dev/null
cp /tmp/prof_stats/bench_kernel_stats.csv $out/bench_kernel_stats.csv 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/prof_fetch -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-e2e --no-dense --no-cpu-baseline > /dev/null 2>&1
cp /tmp/prof_fetch/bench_counter_collection.csv $out/bench_pmc_fetch.csv 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/prof_write -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-e2e --no-dense --no-cpu-baseline > /dev/null 2>&1
cp /tmp/prof_write/bench_counter_collection.csv $out/bench_pmc_write.csv 2>/dev/null
# instruction issue of the same kernels (SQ block, 8 slots; GRBM independent): VALU instructions, VALU-busy quad-cycles, wave quad-cycles,
# where the waves wait
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d /tmp/prof_valu -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-e2e --no-dense --no-cpu-baseline > /dev/null 2>&1
cp /tmp/prof_valu/bench_counter_collection.csv $out/bench_pmc_valu.csv 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS --output-format csv -d /tmp/prof_mix -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-e2e --no-dense --no-cpu-baseline > /dev/null 2>&1
cp /tmp/prof_mix/bench_counter_collection.csv $out/bench_pmc_mix.csv 2>/dev/null
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u > $out/sq_counters.txt
if [ -z "$VFT_PROFILE_OPS" ]; then ls -la $out; cut -c1-400 $out/bench_line.json; exit 0; fi
for a in nt aa; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ops_$a -o ops -- python3 $GRAFT_REPO_ROOT/tools/bench_ops.py $a > $out/ops_table_$a.txt 2> /dev/null
  cp /tmp/prof_ops_$a/ops_kernel_stats.csv $out/ops_${a}_kernel_stats.csv 2>/dev/null
done
ls -la $out
cat $out/bench_line.json | cut -c1-400
