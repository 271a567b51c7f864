#!/bin/bash
# round-6 evidence run on one box: GPU suite, default bench line, steady tables of configs 1 / 2 / 4, PMC passes, launch census, grouping / scatter timings
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|^FAILED|^ERROR" | head -10 > gpurun_out/r06_gpu_suite.txt
python bench.py > gpurun_out/r06_final_bench.json 2> gpurun_out/r06_final_bench.err
bash tools/profile_step.sh r06_fp32_c1 > gpurun_out/r06_profile_c1.log 2>&1
BENCH_ARGS="--config 2" bash tools/profile_step.sh r06_bf16_c2 > gpurun_out/r06_profile_c2.log 2>&1
BENCH_ARGS="--config 4" bash tools/profile_step.sh r06_bf16_c4 > gpurun_out/r06_profile_c4.log 2>&1
bash tools/pmc_step.sh r06 > gpurun_out/r06_pmc_step.log 2>&1
python tools/launch_census.py > gpurun_out/r06_launch_census.txt 2>&1
{ echo "# tools/time_sa_l1.py (i2p_sa_l1_group) and tools/time_fcsk.py (i2p_fused_conv_select_k), level-1 shape, batch 8: us, fraction of 8 TB/s"; python tools/time_sa_l1.py 2>&1 | tail -3; python tools/time_fcsk.py 2>&1 | tail -3; echo "# tools/time_fused_bwd.py (one-pass backward incl. slab reduction; last line with I2P_NO_FUSED_BWD2=1 = the two-kernel form)"; python tools/time_fused_bwd.py 2>&1 | tail -3; I2P_NO_FUSED_BWD2=1 python tools/time_fused_bwd.py 2>&1 | tail -1; } > gpurun_out/r06_grouping.txt
F="--no-cpu-baseline --no-dp-proxy --loader-line 0 --other-configs 0 --steps 40 --warmup 10"
{ echo "# same-box A/B of the deferred weight-gradient reductions (I2P_NO_DEFER=1 = one reduction launch per layer), bench.py $F"; for c in 1 2 4; do for i in 1 2; do
python bench.py $F --config $c 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config $c deferred ', l['value'], l['ms_per_step'])"
I2P_NO_DEFER=1 python bench.py $F --config $c 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config $c immediate', l['value'], l['ms_per_step'])"
done; done; } > gpurun_out/r06_defer_ab.txt
{ echo "# same-box A/B of the two-stream step (image encoder on a second HIP stream; I2P_ONE_STREAM=1 = one stream, encoder chains on), bench.py $F --steps 200 --warmup 20"; for c in 1 2 4; do for i in 1 2; do
python bench.py $F --steps 200 --warmup 20 --config $c 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config $c two streams', l['value'], l['ms_per_step'])"
I2P_ONE_STREAM=1 python bench.py $F --steps 200 --warmup 20 --config $c 2>/dev/null | python -c "import sys,json; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('config $c one stream ', l['value'], l['ms_per_step'])"
done; done; } > gpurun_out/r06_two_stream_ab.txt
{ echo "# tools/time_fwd_streams.py: captured forward pass at configs[1]'s shapes, no profiler"; python tools/time_fwd_streams.py 2>&1 | tail -4
echo "# tools/graph_branch_probe.py: a two-branch hipGraph of one-block spin kernels (A on a second stream, B on the capturing stream), replay time"; python tools/graph_branch_probe.py 2>&1 | tail -17; } > gpurun_out/r06_graph_branches.txt
cat gpurun_out/r06_gpu_suite.txt gpurun_out/r06_defer_ab.txt gpurun_out/r06_two_stream_ab.txt; head -c 300 gpurun_out/r06_final_bench.json
