#!/bin/bash
# Functional check of bench.py's N > 1 path (shards, MaskGather, max-over-ranks timing, JSON) with 2 ranks on ONE GPU:
# both ranks share device 0 and the collectives use gloo.  The numbers it prints are not measurements.
cd "$(dirname "$0")/.."
MPFMT_BENCH_ONE_DEVICE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node ${NPROC:-2} --master-addr 127.0.0.1 --master-port 29577 \
    bench.py --gpus ${NPROC:-2} --steps 3 --warmup 2 --no-cpu-baseline "$@"
