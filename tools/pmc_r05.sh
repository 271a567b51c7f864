#!/bin/bash
# every PMC pass the bench line and DESIGN.md cite, on the final code of the round (results under gpurun_out/<tag>_pmc_*; copy into profiles/):
#   fp32 step kernels  -> <tag>_pmc_traffic.json      (tools/pmc_step.sh: FETCH_SIZE / WRITE_SIZE passes over the in-step kernels)
#   bf16 cost-volume kernels at batch 16 -> <tag>_pmc_bf16_traffic.json (tools/pmc_kernels.sh over tools/time_bf16_bwd.py, calibration pass first)
tag=${1:-r05}
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
bash tools/pmc_step.sh $tag > gpurun_out/${tag}_pmc_step.log 2>&1
tools/pmc_kernels.sh ${tag}_bf16_cal "bwd_stats_bf16_kernel|pair_fwd3_bf16_kernel" python tools/time_bf16_bwd.py --only cal > /dev/null 2>&1
tools/pmc_kernels.sh ${tag}_bf16 "bwd_fused|pair_bwd2|pair_fwd3|rg_fwd_kernel|sm_bwd_bf16|sm_fwd_bf16" python tools/time_bf16_bwd.py --only lin,2src,pair,fwd,sm > /dev/null 2>&1
python tools/pmc_r05_bf16.py gpurun_out/${tag}_bf16_cal_pmc.txt gpurun_out/${tag}_bf16_pmc.txt > gpurun_out/${tag}_pmc_bf16_traffic.json
cat gpurun_out/${tag}_pmc_bf16_traffic.json
ls gpurun_out | grep ${tag}
