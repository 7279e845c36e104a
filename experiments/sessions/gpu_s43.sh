#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/s43; rm -rf $O; mkdir -p $O
cd $R; export PYTHONPATH=$R
( timeout 300 python tools/exp_stagger_single_round.py ) > $O/exp_stagger_single_round.log 2>&1
cat $O/exp_stagger_single_round.log
