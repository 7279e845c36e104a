#!/bin/bash
# round 4, session 1: the four-waves-per-SIMD grouped-wavelength kernels (fft_spectral2.h) against the loop and round 3's groups of 8;
# their kernel trace; the spectral / polychromatic GPU tests on the new default; SQ counters of the headline's two kernels.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/s17; rm -rf $O; mkdir -p $O
cd $R
( timeout 600 python tools/exp_spectral2.py 4096 2048 1024 ) > $O/exp_spectral2.log 2>&1
( timeout 900 python -m pytest tests -x -q -m gpu -k "spectral or config5 or polychrom or poly" 2>&1 | tail -15 ) > $O/pytest_spectral.log 2>&1
( cd /tmp && PROF=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_sp2 -- python $R/tools/exp_spectral2.py 4096 ) > $O/rocprof_sp2.log 2>&1
cp "$(ls $O/prof_sp2/*/*kernel_stats.csv | tail -1)" $O/sp2_kernel_stats.csv
( cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES --output-format csv -d $O/pmc_sq1 -- python $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-poly ) > $O/rocprof_sq1.log 2>&1
( cd /tmp && timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS --output-format csv -d $O/pmc_sq2 -- python $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-poly ) > $O/rocprof_sq2.log 2>&1
python tools/pmc_counters.py $O/pmc_sq1 fft_kernel > $O/headline_sq_counters.txt 2>&1
python tools/pmc_counters.py $O/pmc_sq2 fft_kernel >> $O/headline_sq_counters.txt 2>&1
rm -rf $O/pmc_sq1 $O/pmc_sq2 $O/prof_sp2
cat $O/exp_spectral2.log; tail -5 $O/pytest_spectral.log; head -12 $O/sp2_kernel_stats.csv | cut -c1-200; cat $O/headline_sq_counters.txt | head -60
