#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/s13; rm -rf $O; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_round3.py -x -q -m gpu -k "composite or mixed_radix" 2>&1 | tail -6 ) > $O/pytest_mix.log 2>&1
( PYTHONPATH=$R timeout 600 python tools/exp_mix.py 2>&1 | grep -E "MIX|Error|error" ) > $O/exp_mix.log 2>&1
tail -3 $O/pytest_mix.log; cat $O/exp_mix.log
