#!/bin/bash
# Experiment script (GPU): every bench workload once on one box (round 3: a step is one pass over the orbit's 8 views): value
# (4 frames in flight; path tracing 2), the same frames one at a time, the roofline figures.
# Output: gpurun_out/r3/workloads.txt (one summary line per workload) + the JSON lines.
set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r3/workloads
mkdir -p $O
: > gpurun_out/r3/workloads.txt
for wl in sphere_field_10k_1080p_primary_shadow sphere_field_10k_analytic_1080p_primary_shadow soup_1m_1080p_primary_shadow soup_1m_1080p_4spp soup_1m_4k_path_4spp_4seg soup_10m_1080p_path_1spp_8seg default_640x360; do
  case $wl in *path*) steps=2;; *4spp) steps=6;; *) steps=25;; esac
  timeout -k 10 300 python bench.py --workload $wl --steps $steps --warmup 1 --no-cpu-baseline --no-north-star --no-same-view > $O/$wl.json 2> $O/$wl.err || { tail -3 $O/$wl.err; exit 1; }
  python - $O/$wl.json >> gpurun_out/r3/workloads.txt <<PY
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]; o = d.get("one_frame_at_a_time", {})
print(d["config"]["workload"], "| in flight", d["config"]["frames_in_flight"], "| frames", d["steps"] * d["config"].get("frames_per_step", 1), "| Mrays/s", d["value"], "| ms/frame", d.get("ms_per_frame"),
      "| one at a time: Mrays/s", o.get("value"), "ms", o.get("ms_per_frame"), "| kernel_ms", r["kernel_ms"], "| B/ray", r["bytes_per_ray"], "| frac", r["frac"],
      "| gather frac", (r.get("gather") or {}).get("frac"))
PY
done
cat gpurun_out/r3/workloads.txt
