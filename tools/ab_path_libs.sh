#!/bin/bash
# Experiment script (GPU): builds of the library on the path-tracing workloads.  tools/ab_path_libs.sh lib1.so lib2.so ... ("default" = shipped)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do for lib in "$@"; do for wl in soup_10m_1080p_path_1spp_8seg soup_1m_4k_path_4spp_4seg; do
  if [ "$lib" = default ]; then unset RT_AMD_LIB; else export RT_AMD_LIB=$PWD/$lib; fi
  timeout -k 10 200 python bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-north-star --no-same-view 2>/dev/null | \
   python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('%-28s %-34s' % ('$wl'[:28], '$lib'[-34:]), 'value %8.1f  one-at-a-time %8.1f' % (d['value'], d.get('one_frame_at_a_time',{}).get('value',0)))"
done; done; done
