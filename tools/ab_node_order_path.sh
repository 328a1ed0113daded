#!/bin/bash
# Experiment script (GPU): the wide nodes' memory order (RT_TUNE_WIDE_ORDER 0-4, csrc/bvh_build.cpp wide_reorder) on the workload whose
# traversal is bound by gathers beyond the L2: the 10 M-triangle path workload (and BASELINE configs[2] beside it).
cd ${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
for o in 0 1 2 3 4; do
  for wl in soup_10m_1080p_path_1spp_8seg; do
    RT_TUNE_WIDE_ORDER=$o timeout -k 10 200 python bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-north-star --no-same-view 2>/dev/null | \
      python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('order $o %-36s' % d['config']['workload'], 'value %8.1f  one-at-a-time %8.1f  nodes %d  visits/frame %d tests %d' % (d['value'], d.get('one_frame_at_a_time',{}).get('value',0), d['config']['bvh_nodes'], d['roofline']['node_visits'], d['roofline']['tri_tests']))"
  done
done
done
