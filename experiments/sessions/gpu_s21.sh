#!/bin/bash
# round 4, session 5: config 2 with two units per workgroup (experiment build), grouped-wavelength kernels once more on the experiment build (kept for the record)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/s21; rm -rf $O; mkdir -p $O
cd $R; export PYTHONPATH=$R
( PRYSM_AMD_LIB=$R/prysm_amd/alt/libprysm_amd.so timeout 600 python tools/exp_cfg2_two_units.py ) > $O/exp_cfg2_two_units.log 2>&1
cat $O/exp_cfg2_two_units.log
