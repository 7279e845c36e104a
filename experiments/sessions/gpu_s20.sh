#!/bin/bash
# round 4, session 4: round-4 tests (synthesis in the mixed-radix rows, composites above 8192), composite bench entries, config 2 two streams
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/s20; rm -rf $O; mkdir -p $O
cd $R; export PYTHONPATH=$R
( timeout 1200 python -m pytest tests/test_gpu_round4.py -x -q 2>&1 | tail -25 ) > $O/pytest_round4.log 2>&1
( timeout 600 python bench.py --only composite ) > $O/bench_composite.log 2>&1
( timeout 300 python bench.py --only config2 ) > $O/bench_config2.log 2>&1
( timeout 300 python bench.py --only config3 ) > $O/bench_config3.log 2>&1
tail -12 $O/pytest_round4.log; tail -1 $O/bench_composite.log | python -c "
import json,sys
for k,v in json.loads(sys.stdin.read()).items(): print(k, {a:(round(b,4) if isinstance(b,float) else b) for a,b in v.items() if a!='note'})"; tail -1 $O/bench_config2.log | cut -c1-900; tail -1 $O/bench_config3.log | cut -c1-600
