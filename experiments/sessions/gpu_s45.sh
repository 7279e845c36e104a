#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/s45; rm -rf $O; mkdir -p $O
cd $R; export PYTHONPATH=$R
( timeout 900 python tools/fuzz_fft2.py 500 606 ) > $O/fuzz_500_seed606.log 2>&1
tail -6 $O/fuzz_500_seed606.log
( timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-poly ) > $O/bench20.log 2>&1
tail -1 $O/bench20.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('steps20', d['value'], d['ms_per_step'])"
