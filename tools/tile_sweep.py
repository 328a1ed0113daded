"""Experiment script (GPU): k_stream's work-tile edge (rt_render_params.tile_size; 64 by default). Not a test."""
import os, sys
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_ROOT, os.path.join(_ROOT, "oracle")]
import numpy as np, ray_tracing_amd as rt
for kind in ("field", "soup"):
    app = rt.App()
    if kind == "field": app.spawn_sphere_field(10000, 0x5EED0002, 5)
    else: app.spawn_icosphere_soup(1389, 0x5EED0003, 5)
    app.camera_frame_scene(1.0); app.camera_set_target(1920, 1080)
    sc = app.prepare_meshinfo(); m, p = app.camera_view()
    ctx = rt.Context(0); ctx.upload_scene(sc); ctx.set_view(m, p, 1920, 1080)
    for tile in (16, 32, 64, 128, 256):
        f = rt.FLAG_SHADOW
        ctx.render(flags=f, tile_size=tile); c = ctx.counters()
        for _ in range(3): ctx.render(flags=f, tile_size=tile)
        ctx.counters()
        for _ in range(20): ctx.render(flags=f | rt.FLAG_TIME_KERNELS, tile_size=tile)
        k = ctx.counters(); ms = k.kernel_ms / k.kernel_launches; rays = c.rays_primary + c.rays_shadow
        print(kind, "tile", tile, "ms %.4f Mrays/s %.0f" % (ms, rays / ms / 1e3), flush=True)
    ctx.close()
