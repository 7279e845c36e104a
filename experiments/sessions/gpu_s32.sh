#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/s32; rm -rf $O; mkdir -p $O
cd $R; export PYTHONPATH=$R
( timeout 300 python tools/exp_mix_stagger.py ) > $O/exp_mix_stagger.log 2>&1
cat $O/exp_mix_stagger.log
( timeout 300 python bench.py --only composite ) > $O/bench_composite.log 2>&1
tail -1 $O/bench_composite.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
for k,v in d['other_configs'].items():
    print(k, {a:(round(b,4) if isinstance(b,float) else b) for a,b in v.items() if a in ('ms','composed_ms','frac_of_hbm_peak')})
"
