#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/s33; rm -rf $O; mkdir -p $O
cd $R; export PYTHONPATH=$R
( timeout 900 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1
tail -3 $O/pytest_gpu.log
( timeout 600 python tools/fuzz_fft2.py 400 505 ) > $O/fuzz.log 2>&1
tail -3 $O/fuzz.log
( timeout 300 python tools/exp_mix_pad.py ) > $O/exp_mix_pad.log 2>&1
cat $O/exp_mix_pad.log
( timeout 300 python bench.py --only composite ) > $O/bench_composite.log 2>&1
tail -1 $O/bench_composite.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
for k,v in d.items():
    if isinstance(v, dict): print(k, {a:(round(b,4) if isinstance(b,float) else b) for a,b in v.items() if a in ('ms','composed_ms','frac_of_hbm_peak')})
"
