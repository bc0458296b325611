#!/bin/bash
# round 6, on the GPU box: what the round driver runs at the end - smoke(), the default `pytest -m gpu` suite
out=$GRAFT_REPO_ROOT/gpurun_out/r06suite
mkdir -p $out
cd $GRAFT_REPO_ROOT
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.txt 2>&1; tail -2 $out/smoke.txt
timeout 3300 python3 -m pytest tests -m gpu -x -q --durations=25 > $out/pytest_all.txt 2>&1
tail -45 $out/pytest_all.txt
