#!/bin/bash
# round 4, session 6: LDS padding of the mixed-radix kernels: parity (composite tests + fuzz of the mixed routes), A/B timing, bank-conflict counters
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/s23; rm -rf $O; mkdir -p $O
cd $R; export PYTHONPATH=$R
( timeout 1200 python -m pytest tests -x -q -m gpu -k "composite or mixed or mix or round4 or fuzz or fft1 or reference" 2>&1 | tail -12 ) > $O/pytest_mix.log 2>&1
( timeout 600 python tools/exp_mix_pad.py ) > $O/exp_mix_pad.log 2>&1
for pads in 1 0; do
( cd /tmp && PROF=1 PADS=$pads timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES --output-format csv -d $O/pmc_pad$pads -- python $R/tools/exp_mix_pad.py ) > $O/rocprof_pad$pads.log 2>&1
python tools/pmc_counters.py $O/pmc_pad$pads mix_ > $O/mix_lds_counters_pad$pads.txt 2>&1
rm -rf $O/pmc_pad$pads
done
tail -5 $O/pytest_mix.log; cat $O/exp_mix_pad.log; grep -A9 "mix_" $O/mix_lds_counters_pad1.txt | head -60
