"""Experiment script (GPU): how a primary + shadow frame splits into its two phases (time, wave trips, lane statistics). Not a test."""
import os, sys
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_ROOT]
import ray_tracing_amd as rt
for kind in ("field", "soup"):
    app = rt.App()
    if kind == "field": app.spawn_sphere_field(10000, 0x5EED0002, 5)
    else: app.spawn_icosphere_soup(1389, 0x5EED0003, 5)
    app.camera_frame_scene(1.0); app.camera_set_target(1920, 1080)
    sc = app.prepare_meshinfo(); m, p = app.camera_view()
    ctx = rt.Context(0); ctx.upload_scene(sc); ctx.set_view(m, p, 1920, 1080)
    for name, f in (("primary only", 0), ("primary + shadow", rt.FLAG_SHADOW)):
        ctx.render(flags=f | rt.FLAG_COUNTERS); c = ctx.counters()
        for _ in range(3): ctx.render(flags=f)
        ctx.counters()
        for _ in range(20): ctx.render(flags=f | rt.FLAG_TIME_KERNELS)
        k = ctx.counters(); ms = k.kernel_ms / k.kernel_launches; rays = c.rays_primary + c.rays_shadow
        print(kind, name, "ms %.4f rays %d nodes/ray %.1f tris/ray %.2f alive %.2f wave trips node %d tri %d" % (
            ms, rays, c.node_visits / rays, c.tri_tests / rays, c.wave_alive_lanes / max(1, 64 * c.wave_node_iters), c.wave_node_iters, c.wave_tri_iters), flush=True)
    ctx.close()
