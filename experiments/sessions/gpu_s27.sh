#!/bin/bash
# lean column store of the mixed-radix kernels (A/B: product library at HEAD against the numbers of the round session), and the ablation
# study of the same kernels (experiment build)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/s27; rm -rf $O; mkdir -p $O
cd $R; export PYTHONPATH=$R
( timeout 600 python -m pytest tests -m gpu -x -q -k "mix or composite or primes or fuzz" ) > $O/pytest_mix.log 2>&1
tail -3 $O/pytest_mix.log
( timeout 300 python tools/exp_mix_pad.py ) > $O/exp_mix_pad.log 2>&1
cat $O/exp_mix_pad.log
export PRYSM_AMD_LIB=$R/prysm_amd/alt/libprysm_amd.so
for a in 0 1 2 4 8 3 7 15; do
  ( cd /tmp && ABL=$a timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$a -- python $R/tools/exp_mix_ablate.py ) > $O/rocprof_$a.log 2>&1
  f=$(find $O/prof_$a -name '*kernel_stats.csv' | head -1)
  echo "== ablate $a" >> $O/exp_mix_ablate.log
  python - "$f" >> $O/exp_mix_ablate.log <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'mix_' in r['Name']:
        print('   %-60s calls %4s  avg %8.1f us' % (r['Name'][:60], r['Calls'], float(r['AverageNs']) / 1e3))
PY
done
# instruction counts and issue of the same kernels (product kernels: ablate 0)
( cd /tmp && ABL=0 timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $O/pmc_a -- python $R/tools/exp_mix_ablate.py ) > $O/rocprof_pmc_a.log 2>&1
( cd /tmp && ABL=0 timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES --output-format csv -d $O/pmc_b -- python $R/tools/exp_mix_ablate.py ) > $O/rocprof_pmc_b.log 2>&1
( python tools/pmc_counters.py $O/pmc_a mix_; python tools/pmc_counters.py $O/pmc_b mix_ ) > $O/mix_sq_counters.txt 2>&1
cat $O/exp_mix_ablate.log; cut -c1-150 $O/mix_sq_counters.txt | head -90
find $O -name '*.db' -delete; find $O -name '*_agent_info.csv' -delete
