python -m pytest tests/test_wreg_gpu.py -q 2>&1 | tail -4 > gpurun_out/r06_fused2.log
python -m pytest tests/test_trajectory.py tests/test_model_sized.py -q -m gpu 2>&1 | tail -4 >> gpurun_out/r06_fused2.log
F="--no-cpu-baseline --no-dp-proxy --loader-line 0 --other-configs 0 --steps 40 --warmup 10"
for i in 1 2; do
python bench.py $F 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fused2', l['value'], l['ms_per_step'])" >> gpurun_out/r06_fused2.log
I2P_NO_FUSED_BWD2=1 python bench.py $F 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('two-kernel', l['value'], l['ms_per_step'])" >> gpurun_out/r06_fused2.log
done
cat gpurun_out/r06_fused2.log
