#!/bin/bash
# round 4, session 2: grouped-wavelength kernels (timings + kernel trace), the round-4 GPU tests, two-stream throughput, headline line
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/s18; rm -rf $O; mkdir -p $O
cd $R; export PYTHONPATH=$R
( timeout 600 python tools/exp_spectral2.py 4096 2048 1024 ) > $O/exp_spectral2.log 2>&1
( timeout 900 python -m pytest tests/test_gpu_round4.py -x -q 2>&1 | tail -15 ) > $O/pytest_round4.log 2>&1
( cd /tmp && PROF=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_sp2 -- python $R/tools/exp_spectral2.py 4096 ) > $O/rocprof_sp2.log 2>&1
cp "$(ls $O/prof_sp2/*/*kernel_stats.csv | tail -1)" $O/sp2_kernel_stats.csv; rm -rf $O/prof_sp2
( timeout 300 python tools/exp_two_streams.py 4096; timeout 200 python tools/exp_two_streams.py 2048 ) > $O/exp_two_streams.log 2>&1
( timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline ) > $O/bench20.log 2>&1
cat $O/exp_spectral2.log; tail -5 $O/pytest_round4.log; head -14 $O/sp2_kernel_stats.csv | cut -c1-220; cat $O/exp_two_streams.log; tail -1 $O/bench20.log | cut -c1-1500
