#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/s39; rm -rf $O; mkdir -p $O
cd $R; export PYTHONPATH=$R
( timeout 900 python tools/exp_ab_libs.py prysm_amd/alt/libprysm_amd.so prysm_amd/libprysm_amd.so 2 ) > $O/exp_ab_twpow.log 2>&1
cat $O/exp_ab_twpow.log
