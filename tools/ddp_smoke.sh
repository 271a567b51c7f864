#!/bin/bash
# single-node smoke of the distributed bench path with 1 rank under torchrun (graph A + all-reduce + graph B)
export I2P_FORCE_DP=1
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline
