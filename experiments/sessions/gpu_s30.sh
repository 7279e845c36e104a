#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/s30; rm -rf $O; mkdir -p $O
cd $R; export PYTHONPATH=$R
( timeout 900 python tools/exp_mix_shape.py ) > $O/exp_mix_shape.log 2>&1
cat $O/exp_mix_shape.log
