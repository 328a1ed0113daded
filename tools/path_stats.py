"""Experiment script (GPU): lane utilisation of the path-tracing kernels. Not a test."""
import os, sys
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_ROOT]
import numpy as np, ray_tracing_amd as rt
for name, n, seed, W, H, segs in (("soup1M_4K", 1389, 0x5EED0003, 3840, 2160, 4), ("soup10M_1080p", 13889, 0x5EED0005, 1920, 1080, 8)):
    app = rt.App(); app.spawn_icosphere_soup(n, seed, 5); app.camera_frame_scene(1.0); app.camera_set_target(W, H)
    sc = app.prepare_meshinfo(); m, p = app.camera_view()
    ctx = rt.Context(0); ctx.upload_scene(sc); ctx.set_view(m, p, W, H)
    for s in range(1, segs + 1):
        ctx.render(flags=rt.FLAG_PATH_TRACE | rt.FLAG_COUNTERS, spp=1, max_segments=s, seed=1); c = ctx.counters()
        rays = c.rays_primary + c.rays_bounce
        print(name, "segments<=%d rays %d nodes/ray %.1f tris/ray %.2f alive %.2f util node %.2f tri %.2f wave trips node %d tri %d" % (
            s, rays, c.node_visits / rays, c.tri_tests / rays, c.wave_alive_lanes / max(1, 64 * c.wave_node_iters),
            c.node_visits / max(1, 64 * c.wave_node_iters), c.tri_tests / max(1, 64 * c.wave_tri_iters), c.wave_node_iters, c.wave_tri_iters), flush=True)
    for _ in range(2): ctx.render(flags=rt.FLAG_PATH_TRACE, spp=1, max_segments=segs, seed=1)
    ctx.counters()
    for _ in range(5): ctx.render(flags=rt.FLAG_PATH_TRACE | rt.FLAG_TIME_KERNELS, spp=1, max_segments=segs, seed=1)
    k = ctx.counters(); rays = k.rays_primary + k.rays_bounce
    print(name, "ms/frame %.3f Mrays/s %.0f" % (k.kernel_ms / 5, rays / (k.kernel_ms / 5) / 1e3), flush=True)
    ctx.close()
