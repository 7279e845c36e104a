#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/s49; rm -rf $O; mkdir -p $O
cd $R; export PYTHONPATH=$R
( timeout 600 python -m pytest tests/test_gpu_round4.py -m gpu -x -q ) > $O/pytest_round4.log 2>&1
tail -3 $O/pytest_round4.log
