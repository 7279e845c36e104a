#!/bin/bash
# One GPU-box session for the round's judged artefacts: parity tests, smoke, the default bench line, rocprofv3 kernel stats and
# PMC traffic (FETCH_SIZE / WRITE_SIZE in separate runs) for the headline loop and for every other BASELINE config.  Output ->
# gpurun_out/round/ ; tools/save_profiles.sh copies the summaries into profiles/<round>/.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/round; rm -rf $O; mkdir -p $O
( timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 ) > $O/pytest_gpu.log 2>&1
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/smoke.log 2>&1
( timeout 600 python tools/fuzz_fft2.py ${FUZZ_CASES:-400} ${FUZZ_SEED:-707} 2>&1 | tail -5 ) > $O/fuzz.log 2>&1
# the driver's form of the bench (20 steps) next to the default one
( timeout 600 python bench.py --steps 20 --warmup 5 --no-poly --no-cpu-baseline ) > $O/bench_steps20.log 2>&1
prof() {   # name, bench arguments...
  n=$1; shift
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$n -- python $R/bench.py "$@" ) > $O/rocprof_$n.log 2>&1
  ( cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$n -- python $R/bench.py "$@" ) > $O/rocprof_pmcf_$n.log 2>&1
  ( cd /tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$n -- python $R/bench.py "$@" ) > $O/rocprof_pmcw_$n.log 2>&1
  python tools/pmc_summary.py $O/pmc_fetch_$n $O/pmc_write_$n $O/pmc_${n}_summary.json "bench.py $*" > $O/pmc_${n}_summary.txt 2>&1
  cp "$(ls $O/prof_$n/*/*kernel_stats.csv | tail -1)" $O/${n}_kernel_stats.csv
  rm -rf $O/prof_$n $O/pmc_fetch_$n $O/pmc_write_$n      # the raw traces: gpurun copies at most 64 MiB back
}
prof bench --steps 40 --warmup 5 --no-cpu-baseline --no-poly
# the default bench line AFTER the PMC passes of the same sources: its roofline.traffic comes from the summary just collected
cp $O/pmc_bench_summary.json $R/profiles/pmc_bench_summary.json
( timeout 900 python bench.py ) > $O/bench.log 2>&1
for c in model7 config2 config3 config4 c128 n8192 padded composite mtf conv adjoint poly2048; do prof $c --only $c; done
( cd /tmp && timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/pmc_sq_config4 -- python $R/bench.py --only config4 ) > $O/rocprof_sq_config4.log 2>&1
python tools/pmc_clock.py $O/pmc_sq_config4 $O/pmc_config4_mfma_busy.json 2>&1 | grep pm:: > $O/config4_mfma_busy.txt
cp $O/pmc_config4_mfma_busy.json $R/profiles/pmc_config4_mfma_busy.json
rm -rf $O/pmc_sq_config4
# SQ counters of the headline's two kernels (two passes of eight counters)
( cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES --output-format csv -d $O/pmc_sq1 -- python $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-poly ) > $O/rocprof_sq1.log 2>&1
( cd /tmp && timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $O/pmc_sq2 -- python $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-poly ) > $O/rocprof_sq2.log 2>&1
( python tools/pmc_counters.py $O/pmc_sq1 fft_kernel; python tools/pmc_counters.py $O/pmc_sq2 fft_kernel ) > $O/headline_sq_counters.txt 2>&1
rm -rf $O/pmc_sq1 $O/pmc_sq2
# the N > 1 rehearsal (two ranks share this GPU over gloo): bench.py --gpus 2, the sharded driver against the oracle, the 2-rank image against the 1-rank image
( bash tools/gpu_multi_rank.sh $O/multi_rank ) > $O/multi_rank.log 2>&1
( timeout 300 python tools/exp_graph_branches.py ) > $O/exp_graph_branches.log 2>&1
( timeout 600 python tools/exp_herm_rule.py ) > $O/exp_herm_rule.log 2>&1
( timeout 600 python tools/exp_herm_sweep.py ) > $O/exp_herm_t_log_g.log 2>&1
tail -3 $O/pytest_gpu.log; tail -1 $O/smoke.log; cat $O/multi_rank/check_2rank.json; tail -1 $O/bench.log | cut -c1-400; cat $O/pmc_bench_summary.txt; cat $O/config4_mfma_busy.txt
