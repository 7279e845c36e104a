#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/s13; rm -rf $O; mkdir -p $O
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 ) > $O/smoke.log 2>&1
( timeout 900 python -m pytest tests/test_gpu_round3.py -x -q -m gpu -k "composite" 2>&1 | tail -12 ) > $O/pytest_mix.log 2>&1
cat $O/smoke.log; tail -12 $O/pytest_mix.log
