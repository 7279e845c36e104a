#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/s26; rm -rf $O; mkdir -p $O
cd $R; export PYTHONPATH=$R
( timeout 600 python tools/exp_poly_overlap.py 4096; timeout 300 python tools/exp_poly_overlap.py 2048 ) > $O/exp_poly_overlap.log 2>&1
cat $O/exp_poly_overlap.log
