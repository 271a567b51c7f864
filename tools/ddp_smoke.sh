#!/bin/bash
# single-node smoke of the distributed bench path with 1 rank under torchrun (graph A + all-reduce + graph B)
export I2P_FORCE_DP=1
python bench.py --gpus ${N:-1} --steps 5 --warmup 2 --no-cpu-baseline   # N > 1: bench.py spawns its own ranks
