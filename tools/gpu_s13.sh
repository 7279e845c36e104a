#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/s13; rm -rf $O; mkdir -p $O
( PYTHONPATH=$R timeout 600 python tools/exp_mix_nt.py 2>&1 | grep -E "NT|Error|error" ) > $O/nt.log 2>&1
cat $O/nt.log
