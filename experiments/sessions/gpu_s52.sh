#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/s52; rm -rf $O; mkdir -p $O
cd $R; export PYTHONPATH=$R
( timeout 300 python bench.py --no-cpu-baseline --no-poly ) > $O/bench_headline.log 2>&1
tail -1 $O/bench_headline.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], r['frac'], r['two_plain_copies_ms'], r['step_over_two_plain_copies'], r['traffic'])"
( timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-poly ) > $O/bench20.log 2>&1
tail -1 $O/bench20.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('steps20', d['value'], d['ms_per_step'], r['step_over_two_plain_copies'])"
