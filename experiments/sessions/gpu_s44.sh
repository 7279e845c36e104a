#!/bin/bash
# the experiment build against its own tests (the kernels that are not shipped), then the round session on the product build
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/s44; rm -rf $O; mkdir -p $O
cd $R; export PYTHONPATH=$R
( PRYSM_AMD_LIB=$R/prysm_amd/alt/libprysm_amd.so timeout 900 python -m pytest tests/test_gpu_round4.py -m gpu -x -q ) > $O/pytest_experiment_build.log 2>&1
tail -3 $O/pytest_experiment_build.log
