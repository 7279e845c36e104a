#!/bin/bash
# twiddle powers in the mixed-radix stages (one to three table loads per butterfly instead of R - 1): tests, 2-D times, kernel times
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/s28; rm -rf $O; mkdir -p $O
cd $R; export PYTHONPATH=$R
( timeout 600 python -m pytest tests -m gpu -x -q -k "mix or composite or primes or fuzz or lengths" ) > $O/pytest_mix.log 2>&1
tail -3 $O/pytest_mix.log
( timeout 300 python tools/exp_mix_pad.py ) > $O/exp_mix_pad.log 2>&1
cat $O/exp_mix_pad.log
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $R/tools/exp_mix_ablate.py ) > $O/rocprof.log 2>&1
f=$(find $O/prof -name '*kernel_stats.csv' | head -1)
python - "$f" <<'PY' | tee $O/kernels.log
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'mix_' in r['Name']:
        print('   %-60s calls %4s  avg %8.1f us' % (r['Name'][:60], r['Calls'], float(r['AverageNs']) / 1e3))
PY
find $O -name '*.db' -delete; find $O -name '*_agent_info.csv' -delete
