#!/bin/bash
# rocprofv3 passes over the short fixed workload (tools/pm_gpu_check prof).  Kernel trace + stats first,
# then PMC counters in their OWN runs (never combined with sys/hip traces): FETCH_SIZE and WRITE_SIZE do
# not fit one pass (TCC slots), SQ counters for the MFMA GEMM separately.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
B="$R/tools/pm_gpu_check prof"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $B > $O/trace.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $B > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $B > $O/pmc_write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_sq -- $B > $O/pmc_sq.log 2>&1
find $O -name "*.csv" | head -20; du -sh $O
