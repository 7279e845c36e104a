#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/s42; rm -rf $O; mkdir -p $O
cd $R; export PYTHONPATH=$R
( timeout 300 python tools/exp_fold_small.py ) > $O/exp_fold_small.log 2>&1
cat $O/exp_fold_small.log
