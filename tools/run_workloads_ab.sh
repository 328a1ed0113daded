set -e
for wl in sphere_field_10k_1080p_primary_shadow sphere_field_10k_analytic_1080p_primary_shadow soup_1m_1080p_primary_shadow soup_1m_1080p_4spp soup_1m_4k_path_4spp_4seg soup_10m_1080p_path_1spp_8seg; do
  for fl in 0 0x200; do
    echo "== $wl flags $fl" >> gpurun_out/wl_wide.txt
    timeout -k 10 300 python bench.py --workload $wl --steps 10 --warmup 2 --no-cpu-baseline --no-pipelined --extra-flags $fl >> gpurun_out/wl_wide.txt 2>&1
  done
done
