#!/bin/bash
# One profiling round on the GPU box: GPU tests, bench line, rocprofv3 kernel stats, PMC passes. Output under gpurun_out/.
set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 240 --timeout-method thread > gpurun_out/prof/gpu_tests.txt 2>&1 || { tail -20 gpurun_out/prof/gpu_tests.txt; exit 1; }
tail -1 gpurun_out/prof/gpu_tests.txt
timeout -k 10 300 python bench.py > gpurun_out/prof/bench.json 2> gpurun_out/prof/bench.err || { tail gpurun_out/prof/bench.err; exit 1; }
cat gpurun_out/prof/bench.json
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/stats -- python bench.py --no-cpu-baseline --no-pipelined > gpurun_out/prof/stats.log 2>&1 || { tail gpurun_out/prof/stats.log; exit 1; }
echo stats ok
bash profiles/pmc_passes.sh round && bash profiles/pmc_tcp.sh round_tcp
