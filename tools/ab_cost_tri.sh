#!/bin/bash
# Experiment script (GPU): the SAH triangle cost (leaf size) on the 10 M-triangle path workload and on the bench scene.
# RT_TUNE_COST_TRI: 2 (default: mostly single-triangle leaves) ... 0.25 (leaves of up to RT_TUNE_MAX_LEAF triangles wherever they fit)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for ct in 2 1 0.5 0.25; do
  for ml in 4 8; do
    [ "$ct" = "2" ] && [ "$ml" = "8" ] && continue
    for wl in soup_10m_1080p_path_1spp_8seg sphere_field_10k_1080p_primary_shadow; do
      RT_TUNE_COST_TRI=$ct RT_TUNE_MAX_LEAF=$ml timeout -k 10 200 python bench.py --workload $wl --steps 2 --warmup 1 --no-cpu-baseline --no-north-star --no-same-view 2>/dev/null | \
        python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('cost_tri $ct max_leaf $ml %-40s' % d['config']['workload'], 'value %8.1f  one-at-a-time %8.1f  nodes %d  bytes/ray %.0f  visits/frame %d tests %d' % (d['value'], d.get('one_frame_at_a_time',{}).get('value',0), d['config']['bvh_nodes'], d['roofline']['bytes_per_ray'], d['roofline']['node_visits'], d['roofline']['tri_tests']))"
    done
  done
done
