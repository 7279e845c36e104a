#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/s41; rm -rf $O; mkdir -p $O
cd $R; export PYTHONPATH=$R
( timeout 600 python tools/exp_mix_fold.py ) > $O/exp_mix_fold.log 2>&1
cat $O/exp_mix_fold.log
( timeout 600 python -m pytest tests -m gpu -x -q -k "mix or composite or primes or fuzz or lengths or synth" ) > $O/pytest_mix.log 2>&1
tail -3 $O/pytest_mix.log
