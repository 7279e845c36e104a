#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/s13; rm -rf $O; mkdir -p $O
( timeout 900 python -m pytest tests -x -q -m gpu -k "composite or mixed_radix or bluestein or fft1 or fuzz or angular or conv" 2>&1 | tail -6 ) > $O/pytest_mix.log 2>&1
( PYTHONPATH=$R timeout 600 python tools/exp_mix.py 2>&1 | grep -E "MIX|Error|error" ) > $O/exp_mix.log 2>&1
( cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python $R/bench.py --only composite ) > $O/pmcf.log 2>&1
( cd $R && python tools/pmc_counters.py $O/pmc_fetch mix_ ) > $O/pmc_fetch.txt 2>&1; rm -rf $O/pmc_fetch
tail -5 $O/pytest_mix.log; cat $O/exp_mix.log; cat $O/pmc_fetch.txt
