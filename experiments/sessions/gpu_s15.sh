#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/s15; rm -rf $O; mkdir -p $O; cd /tmp
export PYTHONPATH=$R
B="python $R/tools/exp_mix_prof.py"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $B > $O/trace.log 2>&1
cp "$(ls $O/trace/*/*kernel_stats.csv | tail -1)" $O/mix_kernel_stats.csv
cp "$(ls $O/trace/*/*kernel_trace.csv | tail -1)" $O/mix_kernel_trace.csv
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_sq -- $B > $O/pmc_sq.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_inst -- $B > $O/pmc_inst.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $B > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $B > $O/pmc_write.log 2>&1
cd $R
for d in pmc_sq pmc_inst pmc_fetch pmc_write; do python tools/pmc_counters.py $O/$d mix_ > $O/$d.txt 2>&1; rm -rf $O/$d; done
rm -rf $O/trace
head -12 $O/mix_kernel_stats.csv | cut -c1-160; cat $O/pmc_sq.txt $O/pmc_inst.txt $O/pmc_fetch.txt $O/pmc_write.txt | cut -c1-200
