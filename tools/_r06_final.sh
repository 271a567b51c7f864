#!/bin/bash
# round-6 evidence run on one box: GPU suite, default bench line, steady tables of configs 1 / 2 / 4, grouping / scatter timings
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests -q -m gpu -x 2>&1 | tail -6 > gpurun_out/r06_gpu_suite.txt
python bench.py > gpurun_out/r06_final_bench.json 2> gpurun_out/r06_final_bench.err
bash tools/profile_step.sh r06_fp32_c1 > gpurun_out/r06_profile_c1.log 2>&1
BENCH_ARGS="--config 2" bash tools/profile_step.sh r06_bf16_c2 > gpurun_out/r06_profile_c2.log 2>&1
BENCH_ARGS="--config 4" bash tools/profile_step.sh r06_bf16_c4 > gpurun_out/r06_profile_c4.log 2>&1
{ echo "# tools/time_sa_l1.py (i2p_sa_l1_group) and tools/time_fcsk.py (i2p_fused_conv_select_k), level-1 shape, batch 8: us, fraction of 8 TB/s"; python tools/time_sa_l1.py 2>&1 | tail -3; python tools/time_fcsk.py 2>&1 | tail -3; } > gpurun_out/r06_grouping.txt
cat gpurun_out/r06_gpu_suite.txt; tail -c 600 gpurun_out/r06_final_bench.json
