#!/bin/bash
# exercise bench.py's N > 1 code path on a single GPU: 2 ranks share GPU 0, collectives over gloo
mkdir -p gpurun_out
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 \
    bench.py --gpus 2 --steps 30 --warmup 5 --no-cpu-baseline --backend gloo > gpurun_out/bench_2rank_gloo.log 2>&1
tail -2 gpurun_out/bench_2rank_gloo.log | cut -c1-700
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 \
    tests/multi_rank_poly.py > gpurun_out/poly_2rank_gloo.log 2>&1
grep "polychromatic 2-rank" gpurun_out/poly_2rank_gloo.log || tail -5 gpurun_out/poly_2rank_gloo.log
