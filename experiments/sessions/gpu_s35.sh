#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/s35; rm -rf $O; mkdir -p $O
cd $R; export PYTHONPATH=$R
( timeout 600 python tools/exp_chain_stagger.py ) > $O/exp_chain_stagger.log 2>&1
cat $O/exp_chain_stagger.log
