"""Experiment script (GPU): rays/s of the primary + shadow kernel by frame size (how much of a 1080p frame is ramp-up and tail). Not a test."""
import os, sys
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_ROOT, os.path.join(_ROOT, "oracle")]
import numpy as np, ray_tracing_amd as rt
for kind in ("field", "soup"):
    app = rt.App()
    if kind == "field": app.spawn_sphere_field(10000, 0x5EED0002, 5)
    else: app.spawn_icosphere_soup(1389, 0x5EED0003, 5)
    app.camera_frame_scene(1.0)
    sc = app.prepare_meshinfo()
    ctx = rt.Context(0); ctx.upload_scene(sc)
    for w, h in ((960, 540), (1920, 1080), (2560, 1440), (3840, 2160), (7680, 4320)):
        app.camera_set_target(w, h); m, p = app.camera_view()
        ctx.set_view(m, p, w, h)
        f = rt.FLAG_SHADOW
        ctx.render(flags=f); c = ctx.counters()
        for _ in range(3): ctx.render(flags=f)
        ctx.counters()
        for _ in range(10): ctx.render(flags=f | rt.FLAG_TIME_KERNELS)
        k = ctx.counters(); ms = k.kernel_ms / k.kernel_launches; rays = c.rays_primary + c.rays_shadow
        print(kind, w, h, "ms %.4f Mrays/s %.0f" % (ms, rays / ms / 1e3), flush=True)
    ctx.close()
