#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/s46; rm -rf $O; mkdir -p $O
cd $R; export PYTHONPATH=$R
( timeout 400 python tools/exp_colvar_c128.py ) > $O/exp_colvar_c128.log 2>&1
cat $O/exp_colvar_c128.log
