#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/s50; rm -rf $O; mkdir -p $O
cd $R; export PYTHONPATH=$R
( timeout 900 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1
tail -3 $O/pytest_gpu.log
( python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1
tail -1 $O/smoke.log
