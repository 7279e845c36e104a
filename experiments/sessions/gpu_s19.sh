#!/bin/bash
# round 4, session 3: composite-grid fused chain (tests + bench entries), 1024-thread column kernel sweep, config 2 on two streams, full GPU suite
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/s19; rm -rf $O; mkdir -p $O
cd $R; export PYTHONPATH=$R
( timeout 900 python -m pytest tests/test_gpu_round4.py -x -q 2>&1 | tail -25 ) > $O/pytest_round4.log 2>&1
( timeout 600 python tools/exp_mix_ntc.py ) > $O/exp_mix_ntc.log 2>&1
( timeout 400 python bench.py --only composite ) > $O/bench_composite.log 2>&1
( timeout 300 python bench.py --only config2 ) > $O/bench_config2.log 2>&1
( timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 ) > $O/pytest_gpu.log 2>&1
tail -12 $O/pytest_round4.log; cat $O/exp_mix_ntc.log; tail -1 $O/bench_composite.log | cut -c1-3000; tail -1 $O/bench_config2.log | cut -c1-1200; tail -6 $O/pytest_gpu.log
