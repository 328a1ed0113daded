#!/bin/bash
# One profiling round on the GPU box (round 2): bench line, rocprofv3 kernel stats of the same command, PMC passes for the
# default workload and for north_star's soup scene, pmc_traffic.json regenerated from them.  Output under gpurun_out/prof_r2.
set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
P=gpurun_out/prof_r2
mkdir -p $P
# the default command minus the extras that run the same kernel on another scene; it holds BOTH kinds of launches (4 frames
# in flight in the timed region, then one frame at a time): --stats averages over both, tools/trace_phases.py splits them
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats -- python bench.py --no-cpu-baseline --no-north-star > $P/stats.json 2> $P/stats.log || { tail $P/stats.log; exit 1; }
python tools/trace_phases.py $P/stats "k_stream" > $P/kernel_phases.txt || exit 1
cat $P/kernel_phases.txt
# and one frame at a time only: every launch alone, --stats' average IS the kernel's duration
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats_one -- python bench.py --in-flight 1 --no-cpu-baseline --no-pipelined --no-north-star > $P/stats_one.json 2> $P/stats_one.log || { tail $P/stats_one.log; exit 1; }
echo stats ok
if [ "${1:-}" = "stats-only" ]; then timeout -k 10 300 python bench.py > $P/bench.json 2> $P/bench.err || { tail $P/bench.err; exit 1; }; cat $P/bench.json; exit 0; fi
bash tools/pmc_full.sh r2_field "k_stream<false, true, false>" > $P/pmc_field.log 2>&1 || { tail $P/pmc_field.log; exit 1; }
python profiles/pmc_to_json.py gpurun_out/pmc_r2_field "k_stream<false, true, false>" sphere_field_10k_1080p_primary_shadow "profiles/r2/pmc_k_stream.txt (rocprofv3 --pmc, one run per counter group, tools/pmc_full.sh)" > $P/pmc_json.log 2>&1 || { tail $P/pmc_json.log; exit 1; }
cp profiles/pmc_traffic.json $P/pmc_traffic.json
bash tools/pmc_full.sh r2_soup "k_stream<false, true, false>" --workload soup_1m_1080p_primary_shadow > $P/pmc_soup.log 2>&1 || { tail $P/pmc_soup.log; exit 1; }
timeout -k 10 300 python bench.py > $P/bench.json 2> $P/bench.err || { tail $P/bench.err; exit 1; }
cat $P/bench.json
