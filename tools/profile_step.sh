#!/bin/bash
# rocprofv3 kernel trace of bench.py, reduced to the timed steps.  usage: tools/profile_step.sh <tag>
# (hipGraph capture runs 3 eager steps first; the window is steps 1..4 of the 5 timed ones so that it ends on a step boundary)
tag=${1:-step}
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$tag -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --other-configs 0 --loader-line 0 $BENCH_ARGS > /tmp/prof_$tag.log 2>&1
tail -1 /tmp/prof_$tag.log | cut -c1-200
trace=$(find /tmp/prof_$tag -name '*kernel_trace.csv' | head -1)
mkdir -p gpurun_out
python tools/steady_stats.py "$trace" --warmup 5 --steps 4 --top 70 --out gpurun_out/${tag}_steady_kernel_stats.csv --train-steps 10 --tail-out gpurun_out/${tag}_roofline_section_kernel_stats.csv
python tools/step_sequence.py "$trace" 7 > gpurun_out/${tag}_step_sequence.txt 2>&1
# (two-stream step: under the profiler the second queue of a replayed graph starts late — compare tools/graph_branch_probe.py with and
#  without rocprofv3 — so the timeline shows the overlap of the backward pass but understates that of the forward pass)
python tools/step_timeline.py "$trace" 7 > gpurun_out/${tag}_step_timeline.txt 2>&1
