#!/bin/bash
# Rehearsal of bench.py's N > 1 code path on a ONE-GPU box: 2 ranks share GPU 0, collectives over gloo -- the
# self-launch, the barrier / MAX-over-ranks timing, all three reduce forms, the pipelined frames, the watchdog.  Then the sharded
# polychromatic driver against the oracle (tests/multi_rank_poly.py) and the 2-rank image of config 5 against the 1-rank image
# (tools/poly_image.py + tools/check_2rank.py).  Output -> $O (default gpurun_out/multi_rank).  RCCL with N > 1 needs N devices: it
# stays unmeasured on hardware until a multi-GPU node runs this bench.
R=${GRAFT_REPO_ROOT:-$PWD}; O=${1:-$R/gpurun_out/multi_rank}; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
# (1) the bench itself, launched the way a user would: python bench.py --gpus 2 (self_launch -> torch.distributed.run)
( timeout 900 python $R/bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline --backend gloo ) > $O/bench_2rank_gloo.log 2>&1
grep '"metric"' $O/bench_2rank_gloo.log | tail -1 > $O/bench_2rank_gloo.json
# (2) the sharded driver against the oracle, 2 ranks
( timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 \
    $R/tests/multi_rank_poly.py ) > $O/poly_2rank_gloo.log 2>&1
grep "polychromatic 2-rank" $O/poly_2rank_gloo.log || tail -5 $O/poly_2rank_gloo.log
# (3) config 5's image at 2048^2: 1 rank, then 2 ranks with every reduce form + pipelined frames
( timeout 300 python $R/tools/poly_image.py /tmp/poly_img_1rank.npz 2048 ) > $O/poly_image.log 2>&1
( timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 \
    $R/tools/poly_image.py /tmp/poly_img_2rank.npz 2048 ) >> $O/poly_image.log 2>&1
python $R/tools/check_2rank.py $O/bench_2rank_gloo.json /tmp/poly_img_1rank.npz /tmp/poly_img_2rank.npz | tee $O/check_2rank.json
