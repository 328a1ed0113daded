set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/refresh
timeout -k 10 500 python -u tools/check_8k.py > gpurun_out/refresh/check_8k.txt 2>&1
timeout -k 10 300 python -u tools/refit_time.py > gpurun_out/refresh/refit_time.txt 2>&1
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/refresh/path4k -- python bench.py --workload soup_1m_4k_path_4spp_4seg --steps 5 --warmup 1 --no-cpu-baseline --no-pipelined > gpurun_out/refresh/path4k.log 2>&1
timeout -s KILL 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/refresh/refit -- python tools/refit_time.py > gpurun_out/refresh/refit_prof.log 2>&1
echo ok
