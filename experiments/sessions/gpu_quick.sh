mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 400 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 ) > gpurun_out/pytest_gpu.log 2>&1
( timeout 120 python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/smoke.log 2>&1
( timeout 200 python bench.py --steps 200 --warmup 20 ) > gpurun_out/bench.log 2>&1
tail -3 gpurun_out/pytest_gpu.log; tail -2 gpurun_out/smoke.log; tail -1 gpurun_out/bench.log | cut -c1-400
