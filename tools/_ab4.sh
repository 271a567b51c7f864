#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_two_streams_gpu.py tests/test_train_gpu.py -q -m gpu -x -s 2>&1 | grep -v "^$" | tail -30 > gpurun_out/r06_ts_tests2.txt
bash tools/profile_step.sh r06ts_fp32_c1 > gpurun_out/r06ts_profile_c1.log 2>&1
BENCH_ARGS="--config 2" bash tools/profile_step.sh r06ts_bf16_c2 > gpurun_out/r06ts_profile_c2.log 2>&1
cat gpurun_out/r06_ts_tests2.txt; head -1 gpurun_out/r06ts_*_steady_kernel_stats.csv
