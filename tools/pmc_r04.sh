#!/bin/bash
# every PMC pass the bench line and DESIGN.md cite, on the final code of the round: fp32 step kernels, bf16 roofline kernels, image tail,
# chain kernels.  Results under gpurun_out/<tag>_pmc_*; copy into profiles/.
tag=${1:-r04}
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
bash tools/pmc_step.sh $tag > gpurun_out/${tag}_pmc_step.log 2>&1
bash tools/pmc_traffic.sh bf16 > gpurun_out/${tag}_pmc_bf16.log 2>&1
cp gpurun_out/pmc_bf16_FETCH_SIZE.txt gpurun_out/${tag}_pmc_bf16_FETCH_SIZE.txt; cp gpurun_out/pmc_bf16_WRITE_SIZE.txt gpurun_out/${tag}_pmc_bf16_WRITE_SIZE.txt
python tools/pmc_bf16_json.py gpurun_out/${tag}_pmc_bf16_FETCH_SIZE.txt gpurun_out/${tag}_pmc_bf16_WRITE_SIZE.txt $tag > gpurun_out/${tag}_pmc_bf16_traffic.json
bash tools/pmc_img.sh $tag > gpurun_out/${tag}_pmc_img.log 2>&1
bash tools/pmc_chain.sh $tag > gpurun_out/${tag}_pmc_chain.log 2>&1
ls gpurun_out | grep ${tag}_pmc
