"""Experiment script (GPU): not a test, not collected by pytest."""
import os
import sys
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_ROOT, os.path.join(_ROOT, "oracle")]
import numpy as np, ray_tracing_amd as rt
ctx = rt.Context(0)
for kind,n,seed in (("field",10000,0x5EED0002),("soup",1389,0x5EED0003)):
    app = rt.App()
    (app.spawn_sphere_field if kind=="field" else app.spawn_icosphere_soup)(n, seed, 5)
    sc = app.prepare_meshinfo(); ctx.upload_scene(sc)
    app.camera_set_target(1920,1080)
    for r, lane in ((2.5,0),(1.0,0),(0.4,0)):
        app.camera_frame_scene(r); m,p = app.camera_view(); ctx.set_view(m,p,1920,1080)
        ctx.render(flags=(rt.FLAG_SHADOW|lane)|rt.FLAG_COUNTERS); c=ctx.counters()
        for _ in range(3): ctx.render(flags=(rt.FLAG_SHADOW|lane))
        for _ in range(10): ctx.render(flags=(rt.FLAG_SHADOW|lane)|rt.FLAG_TIME_KERNELS)
        k=ctx.counters(); ms=k.kernel_ms/k.kernel_launches
        rays=c.rays_primary+c.rays_shadow
        print(kind, r, "persist" if lane else "static", "hit %.3f"%(c.rays_shadow/c.rays_primary), "nodes/ray %.1f tris/ray %.2f"%(c.node_visits/rays, c.tri_tests/rays), "ms %.3f Mrays/s %.0f"%(ms, rays/ms/1e3), "alive %.2f util node %.2f tri %.2f witers/wave %d %d"%(c.wave_alive_lanes/max(1,64*c.wave_node_iters), c.node_visits/max(1,64*c.wave_node_iters), c.tri_tests/max(1,64*c.wave_tri_iters), c.wave_node_iters/32640, c.wave_tri_iters/32640), flush=True)
