#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/s40; rm -rf $O; mkdir -p $O
cd $R; export PYTHONPATH=$R
export PRYSM_AMD_LIB=$R/prysm_amd/alt/libprysm_amd.so
( timeout 900 python tools/exp_mix_weights.py ) > $O/exp_mix_weights.log 2>&1
cat $O/exp_mix_weights.log
