#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/s13; rm -rf $O; mkdir -p $O
( cd $R && PYTHONPATH=$R timeout 1500 python tools/fuzz_fft2.py 900 2026 2>&1 | tail -12 ) > $O/fuzz.log 2>&1
cat $O/fuzz.log
