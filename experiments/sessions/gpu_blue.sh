#!/bin/bash
# Bluestein path: its GPU tests, the fuzzer, the timing experiment.  Output -> gpurun_out/
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 300 python -m pytest tests -x -q -m gpu -k "bluestein or fuzz or focus_vs_oracle or rectangular" 2>&1 | tail -25 ) > gpurun_out/pytest_blue.log 2>&1
( timeout 200 python tools/exp_bluestein.py 2>&1 | grep -E "EXP|Error|error" ) > gpurun_out/exp_bluestein.log 2>&1
tail -5 gpurun_out/pytest_blue.log; cat gpurun_out/exp_bluestein.log
