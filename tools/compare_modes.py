"""Experiment script (GPU): not a test, not collected by pytest."""
import os
import sys
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_ROOT, os.path.join(_ROOT, "oracle")]
import numpy as np, ray_tracing_amd as rt
ctx = rt.Context(0)
def run(name, app, W, H):
    sc = app.prepare_meshinfo(); ctx.upload_scene(sc)
    app.camera_set_target(W,H); m,p = app.camera_view(); ctx.set_view(m,p,W,H)
    for mode, fl in (("static",0),("packet",rt.FLAG_WAVE_TRAVERSAL),("persist",rt.FLAG_STATIC_GRID),("brute",rt.FLAG_BRUTE_FORCE)):
        if mode=="brute" and len(sc.triangles)>5000: continue
        f = rt.FLAG_SHADOW|fl
        for _ in range(3): ctx.render(flags=f)
        ctx.counters()
        for _ in range(10): ctx.render(flags=f|rt.FLAG_TIME_KERNELS)
        k=ctx.counters(); ms=k.kernel_ms/k.kernel_launches; rays=k.rays_primary+k.rays_shadow
        print(name, mode, "ms %.4f Mrays/s %.0f"%(ms, rays/ms/1e3), flush=True)
    # path tracing
    for spp,seg in ((4,4),):
        f = rt.FLAG_PATH_TRACE
        ctx.render(flags=f, spp=1, max_segments=seg); ctx.counters()
        ctx.render(flags=f|rt.FLAG_TIME_KERNELS, spp=spp, max_segments=seg, seed=1)
        k=ctx.counters(); rays=k.rays_primary+k.rays_bounce
        print(name, "path %dspp %dseg"%(spp,seg), "ms %.3f Mrays/s %.0f launches %d rays %d"%(k.kernel_ms, rays/k.kernel_ms/1e3, k.kernel_launches, rays), flush=True)
app = rt.App(); app.spawn_default_scene(); run("default1080p", app, 1920, 1080)
app = rt.App(); app.spawn_default_scene(); run("default360p", app, 640, 360)
app = rt.App(); app.spawn_icosphere_soup(1389, 0x5EED0003, 5); app.camera_frame_scene(1.0); run("soup1M_1080p", app, 1920, 1080)
app = rt.App(); app.spawn_icosphere_soup(1389, 0x5EED0003, 5); app.camera_frame_scene(1.0); run("soup1M_4K", app, 3840, 2160)
