#!/bin/bash
# Experiment script (GPU): every bench workload once, 10 steps, on one box: value (4 frames in flight), the same frames one at
# a time, the roofline figures.  Output: gpurun_out/r2/workloads.txt (one summary line per workload) + the JSON lines.
set -u
O=gpurun_out/r2/workloads
mkdir -p $O
: > gpurun_out/r2/workloads.txt
for wl in sphere_field_10k_1080p_primary_shadow sphere_field_10k_analytic_1080p_primary_shadow soup_1m_1080p_primary_shadow soup_1m_1080p_4spp soup_1m_4k_path_4spp_4seg soup_10m_1080p_path_1spp_8seg default_640x360; do
  timeout -k 10 300 python bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline --no-north-star > $O/$wl.json 2> $O/$wl.err || { tail -3 $O/$wl.err; exit 1; }
  python - $O/$wl.json >> gpurun_out/r2/workloads.txt <<PY
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]; o = d.get("one_frame_at_a_time", {})
print(d["config"]["workload"], "| in flight", d["config"]["frames_in_flight"], "| Mrays/s", d["value"], "| ms/frame", d["ms_per_step"],
      "| one at a time: Mrays/s", o.get("value"), "ms", o.get("ms_per_step"), "| kernel_ms", r["kernel_ms"], "| B/ray", r["bytes_per_ray"], "| frac", r["frac"])
PY
done
cat gpurun_out/r2/workloads.txt
