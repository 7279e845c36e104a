#!/bin/bash
# One GPU-box session: parity tests, smoke, bench, rocprof kernel trace.  Output -> gpurun_out/
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 ) > gpurun_out/pytest_gpu.log 2>&1
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/smoke.log 2>&1
( timeout 600 python bench.py --steps 200 --warmup 20 ) > gpurun_out/bench.log 2>&1
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_bench -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline ) > gpurun_out/rocprof_bench.log 2>&1
tail -3 gpurun_out/pytest_gpu.log; cat gpurun_out/smoke.log | tail -2; cat gpurun_out/bench.log | tail -1 | cut -c1-600
find gpurun_out/prof_bench -name "*stats*" | head
