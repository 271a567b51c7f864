#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p gpurun_out
F="--no-cpu-baseline --no-dp-proxy --loader-line 0 --other-configs 0 --steps 200 --warmup 20"
run() { python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', l['value'], l['ms_per_step'])"; }
{ for c in 1 2 4; do
I2P_NO_CHAIN=1 timeout 600 python bench.py $F --config $c 2>gpurun_out/r06_ts3_err.txt | run "config $c no-chain two streams"
I2P_NO_CHAIN=1 I2P_ONE_STREAM=1 timeout 600 python bench.py $F --config $c 2>/dev/null | run "config $c no-chain one stream "
I2P_ONE_STREAM=1 timeout 600 python bench.py $F --config $c 2>/dev/null | run "config $c chain    one stream "
done; } > gpurun_out/r06_ts3_ab.txt 2>&1
cat gpurun_out/r06_ts3_ab.txt; tail -3 gpurun_out/r06_ts3_err.txt
