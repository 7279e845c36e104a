#!/bin/bash
# One gpurun call = one session: `tools/gpu_session.sh <name> <step> [<step> ...]`, output under gpurun_out/<name>/ (copy what is to be
# judged into profiles/<round>/).  Steps (each logs to its own file and is wrapped in `timeout`):
#   tests[:<pytest -k expr>]   the GPU parity suite (or a subset)            -> pytest_gpu.log
#   smoke                      __graft_entry__.smoke()                       -> smoke.log
#   bench[:<bench.py args>]    the bench line                               -> bench*.log
#   ab:<libA>:<libB>[:rounds]  two builds against each other (exp_ab_libs)  -> ab.log
#   knob:<knob>:<v0,v1>:<workload>[,<workload>...][:rounds]                 -> knob_<knob>.log
#   py:<script.py>[:args]      any tools/ experiment script                 -> <script>.log
#   sh:<script.sh>[:args]      a shell script, called with its output directory $O/<script> first -> <script>.log
#   prof:<tag>:<bench.py args> rocprofv3 kernel stats + FETCH / WRITE passes + summary (tools/pmc_summary.py)
#   sq:<tag>:<kernel substr>:<bench.py args>   two SQ counter passes -> <tag>_sq_counters.txt
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; NAME=$1; shift
O=$R/gpurun_out/$NAME; rm -rf $O; mkdir -p $O
export PYTHONPATH=$R:$PYTHONPATH
for step in "$@"; do
  IFS=: read -r kind a b c d <<< "$step"
  case $kind in
    tests) if [ -n "$a" ]; then ( timeout 1500 python -m pytest tests -x -q -m gpu -k "$a" 2>&1 | tail -25 ) > $O/pytest_gpu.log 2>&1; else ( timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 ) > $O/pytest_gpu.log 2>&1; fi; tail -3 $O/pytest_gpu.log ;;
    smoke) ( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1; tail -1 $O/smoke.log ;;
    bench) ( timeout 900 python bench.py $a ) > $O/bench${b:+_$b}.log 2>&1; tail -1 $O/bench${b:+_$b}.log | cut -c1-600 ;;
    ab) ( timeout 900 python tools/exp_ab_libs.py $a $b ${c:-2} ) > $O/ab.log 2>&1; cat $O/ab.log ;;
    knob) ( timeout 600 python tools/exp_knob_ab.py $a $b ${c//,/ } ${d:-3} ) > $O/knob_$a.log 2>&1; cat $O/knob_$a.log ;;
    sh) n=$(basename $a .sh); ( timeout 2400 bash $a $O/$n $b ) > $O/$n.log 2>&1; tail -15 $O/$n.log ;;
    py) n=$(basename $a .py); ( timeout 900 python $a $b ) > $O/$n.log 2>&1; tail -40 $O/$n.log ;;
    prof)
      ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$a -- python $R/bench.py $b ) > $O/rocprof_$a.log 2>&1
      ( cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$a -- python $R/bench.py $b ) > $O/rocprof_pmcf_$a.log 2>&1
      ( cd /tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$a -- python $R/bench.py $b ) > $O/rocprof_pmcw_$a.log 2>&1
      python tools/pmc_summary.py $O/pmc_fetch_$a $O/pmc_write_$a $O/pmc_${a}_summary.json "bench.py $b" > $O/pmc_${a}_summary.txt 2>&1
      cp "$(ls $O/prof_$a/*/*kernel_stats.csv | tail -1)" $O/${a}_kernel_stats.csv; rm -rf $O/prof_$a $O/pmc_fetch_$a $O/pmc_write_$a
      head -8 $O/${a}_kernel_stats.csv | cut -c1-220; cat $O/pmc_${a}_summary.txt ;;
    sq)
      ( cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES --output-format csv -d $O/pmc_sq1 -- python $R/bench.py $c ) > $O/rocprof_sq1.log 2>&1
      ( cd /tmp && timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $O/pmc_sq2 -- python $R/bench.py $c ) > $O/rocprof_sq2.log 2>&1
      ( python tools/pmc_counters.py $O/pmc_sq1 $b; python tools/pmc_counters.py $O/pmc_sq2 $b ) > $O/${a}_sq_counters.txt 2>&1
      rm -rf $O/pmc_sq1 $O/pmc_sq2; cat $O/${a}_sq_counters.txt ;;
    *) echo "unknown step $step" ;;
  esac
done
