#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/s37; rm -rf $O; mkdir -p $O
cd $R; export PYTHONPATH=$R
( timeout 600 python tools/exp_mtf_stagger.py ) > $O/exp_mtf_stagger.log 2>&1
cat $O/exp_mtf_stagger.log
