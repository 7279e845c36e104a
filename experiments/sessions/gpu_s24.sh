#!/bin/bash
# round 4, session 8: the radix-8 engine (8 points per thread) on the headline's two passes, experiment build
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/s24; rm -rf $O; mkdir -p $O
cd $R; export PYTHONPATH=$R
( PRYSM_AMD_LIB=$R/prysm_amd/alt/libprysm_amd.so timeout 600 python tools/exp_engine_p8.py ) > $O/exp_engine_p8.log 2>&1
cat $O/exp_engine_p8.log
