#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/s7; rm -rf $O; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_round3.py -x -q -m gpu 2>&1 | tail -8 ) > $O/pytest_r3.log 2>&1
( timeout 1500 python tools/fuzz_fft2.py 400 3003 2>&1 | tail -12 ) > $O/fuzz_400_seed3003.log 2>&1
tail -4 $O/pytest_r3.log; tail -8 $O/fuzz_400_seed3003.log
