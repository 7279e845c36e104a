#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/s51; rm -rf $O; mkdir -p $O
cd $R; export PYTHONPATH=$R
( timeout 300 python tools/exp_copy_ceiling.py ) > $O/exp_copy_ceiling.log 2>&1
cat $O/exp_copy_ceiling.log
