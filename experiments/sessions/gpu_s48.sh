#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/s48; rm -rf $O; mkdir -p $O
cd $R; export PYTHONPATH=$R
( timeout 300 python tools/exp_colvar3.py ) > $O/exp_colvar3.log 2>&1
cat $O/exp_colvar3.log
