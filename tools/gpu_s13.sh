#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/s13; rm -rf $O; mkdir -p $O
( PYTHONPATH=$R timeout 900 python tools/exp_mix_sweep.py 2>&1 | grep -E "SWEEP|Error|error" ) > $O/sweep.log 2>&1
cat $O/sweep.log
