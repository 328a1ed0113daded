"""Experiment script (GPU): what a path segment costs by its position in the path -- frames with max_segments = 1..8, time and
ray count differences between consecutive settings (10 M-triangle soup, 1080p, 1 spp).  Segment 2 starts at screen-coherent
hit points with random directions; later segments start anywhere.  Not a test."""
import os, sys
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__)))]
import ray_tracing_amd as rt
app = rt.App(); app.spawn_icosphere_soup(13889, 0x5EED0005, 5); app.camera_frame_scene(1.0); app.camera_set_target(1920, 1080)
sc = app.prepare_meshinfo(); m, p = app.camera_view()
ctx = rt.Context(0); ctx.upload_scene(sc); ctx.set_view(m, p, 1920, 1080)
prev = (0.0, 0, 0, 0)
for s in range(1, 9):
    ctx.render(flags=rt.FLAG_PATH_TRACE | rt.FLAG_COUNTERS, spp=1, max_segments=s, seed=1); c = ctx.counters()
    rays, nodes, trips = c.rays_primary + c.rays_bounce, c.node_visits, c.wave_node_iters
    for _ in range(2): ctx.render(flags=rt.FLAG_PATH_TRACE, spp=1, max_segments=s, seed=1)
    ctx.counters()
    for _ in range(6): ctx.render(flags=rt.FLAG_PATH_TRACE | rt.FLAG_TIME_KERNELS, spp=1, max_segments=s, seed=1)
    ms = ctx.counters().kernel_ms / 6
    dr, dn, dt, dw = rays - prev[1], nodes - prev[2], ms - prev[0], trips - prev[3]
    print("segment %d: %8d rays  %.3f ms  %.2f ns/ray  %.0f Mrays/s  nodes/ray %.1f  unfinished lanes per node trip %.1f" % (
        s, dr, dt, dt * 1e6 / dr, dr / dt / 1e3, dn / dr, dn / max(dw, 1)), flush=True)
    prev = (ms, rays, nodes, trips)
ctx.close()
