#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/s13; rm -rf $O; mkdir -p $O
( PYTHONPATH=$R timeout 600 python tools/exp_mix.py 2>&1 | grep -E "MIX|Error|error" ) > $O/exp_mix.log 2>&1
cat $O/exp_mix.log
