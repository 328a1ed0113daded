#!/bin/bash
# The path-tracing workloads of bench.py, 10 steps each (experiments: compare two builds with RT_AMD_LIB).
set -e
for wl in soup_1m_4k_path_4spp_4seg soup_10m_1080p_path_1spp_8seg soup_1m_1080p_4spp; do
  echo "== ${RT_AMD_LIB:-default} $wl"
  timeout -k 10 300 python bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline --no-pipelined 2>&1 | python -c "import sys,json; [print('   ', json.loads(l)['value'], 'Mrays/s', json.loads(l)['ms_per_step'], 'ms') for l in sys.stdin if l.startswith('{')]"
done
