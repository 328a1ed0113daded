#!/bin/bash
# One profiling round on the GPU box (round 3): rocprofv3 kernel stats of the default bench command (and of the same command one
# frame at a time), PMC passes for the kernel that runs alone (k_stream<false, true>), for the kernel of the TIMED region
# (k_stream8<false, true>, forced with RT_TUNE_STREAM_WAVES=8 because a counter pass serialises launches) and for the path
# kernels of configs[2] and of the 10 M-triangle scene; profiles/pmc_traffic.json regenerated from them; the bench line.
# Output under gpurun_out/prof_r3 (copy what is to be judged into profiles/r3/).
set -u
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
P=gpurun_out/prof_r3
mkdir -p $P
# stages (a box call is limited to 20 minutes, and gpurun_out/ does not travel TO the box): a = kernel stats + the k_stream passes,
# b = the path kernels' passes, json = pmc_traffic.json from the merged CSVs (runs anywhere: needs only the CSVs and the
# library for its fingerprint), all = everything in one call
STAGE=${1:-all}
if [ "$STAGE" = "all" ] || [ "$STAGE" = "a" ] || [ "$STAGE" = "stats-only" ]; then
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats -- python bench.py --no-cpu-baseline --no-north-star --no-same-view > $P/stats.json 2> $P/stats.log || { tail $P/stats.log; exit 1; }
python tools/trace_phases.py $P/stats "k_stream" > $P/kernel_phases.txt || exit 1
cat $P/kernel_phases.txt
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats_one -- python bench.py --in-flight 1 --no-cpu-baseline --no-north-star > $P/stats_one.json 2> $P/stats_one.log || { tail $P/stats_one.log; exit 1; }
echo stats ok
if [ "${1:-}" = "stats-only" ]; then exit 0; fi
fi
if [ "$STAGE" = "all" ] || [ "$STAGE" = "a" ]; then
bash tools/pmc_full.sh r3_field "k_stream<false, true>" --in-flight 1 > $P/pmc_field.log 2>&1 || { tail $P/pmc_field.log; exit 1; }
RT_TUNE_STREAM_WAVES=8 bash tools/pmc_full.sh r3_field8 "k_stream8<false, true>" --in-flight 1 > $P/pmc_field8.log 2>&1 || { tail $P/pmc_field8.log; exit 1; }
fi
if [ "$STAGE" = "all" ] || [ "$STAGE" = "b" ]; then
PMC_STEPS=1 bash tools/pmc_full.sh r3_path10m "k_path_start<false>;k_path_bounce<false>" --in-flight 1 --workload soup_10m_1080p_path_1spp_8seg > $P/pmc_path10m.log 2>&1 || { tail $P/pmc_path10m.log; exit 1; }
PMC_STEPS=1 bash tools/pmc_full.sh r3_configs2 "k_path_start<false>" --in-flight 1 --workload soup_1m_1080p_4spp > $P/pmc_configs2.log 2>&1 || { tail $P/pmc_configs2.log; exit 1; }
fi
if [ "$STAGE" = "all" ] || [ "$STAGE" = "json" ]; then
python profiles/pmc_to_json.py sphere_field_10k_1080p_primary_shadow 8 "profiles/r3/pmc_k_stream.txt, pmc_k_stream8.txt, pmc_path_1080p_soup_10m.txt, pmc_path_1080p_soup_1m_4spp.txt (rocprofv3 --pmc, one run per counter group, tools/profile_round3.sh)" \
    main=gpurun_out/pmc_r3_field in_flight=gpurun_out/pmc_r3_field8 path_10m=gpurun_out/pmc_r3_path10m configs2=gpurun_out/pmc_r3_configs2 > $P/pmc_json.log 2>&1 || { tail $P/pmc_json.log; exit 1; }
cp profiles/pmc_traffic.json $P/pmc_traffic.json
for n in r3_field r3_field8 r3_path10m r3_configs2; do cp gpurun_out/pmc_$n/summary.txt $P/pmc_$n.txt; done
fi
if [ "$STAGE" = "all" ] || [ "$STAGE" = "bench" ]; then
timeout -k 10 300 python bench.py > $P/bench.json 2> $P/bench.err || { tail $P/bench.err; exit 1; }
cat $P/bench.json
fi
