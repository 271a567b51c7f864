#!/bin/bash
# same-box A/B: image encoder on a second stream (default) vs I2P_ONE_STREAM=1
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_model_sized.py tests/test_model_golden.py tests/test_train_gpu.py tests/test_trajectory.py -q -m gpu -x 2>&1 | tail -30 > gpurun_out/r06_ts_tests.txt
F="--no-cpu-baseline --no-dp-proxy --loader-line 0 --other-configs 0 --steps 200 --warmup 20"
{ for c in 1 2 4; do for i in 1 2; do
timeout 600 python bench.py $F --config $c 2>gpurun_out/r06_ts_err_$c.txt | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config $c two streams', l['value'], l['ms_per_step'])"
I2P_ONE_STREAM=1 timeout 600 python bench.py $F --config $c 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config $c one stream ', l['value'], l['ms_per_step'])"
done; done; } > gpurun_out/r06_ts_ab.txt 2>&1
cat gpurun_out/r06_ts_tests.txt gpurun_out/r06_ts_ab.txt; tail -3 gpurun_out/r06_ts_err_1.txt
