#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/s34; rm -rf $O; mkdir -p $O
cd $R; export PYTHONPATH=$R
( timeout 600 python tools/exp_fft_stagger.py ) > $O/exp_fft_stagger.log 2>&1
cat $O/exp_fft_stagger.log
