#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/s14; rm -rf $O; mkdir -p $O
( timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -12 ) > $O/pytest.log 2>&1
( PYTHONPATH=$R timeout 900 python tools/fuzz_fft2.py 500 31 2>&1 | tail -25 ) > $O/fuzz.log 2>&1
( PYTHONPATH=$R timeout 600 python tools/exp_mix.py 2>&1 | grep -E "MIX|Error|error" ) > $O/exp_mix.log 2>&1
tail -6 $O/pytest.log; tail -12 $O/fuzz.log; cat $O/exp_mix.log
