"""Experiment script (CPU only): host BVH build time of the bench scene and the 1 M-triangle soup by thread count. Not a test."""
import os, sys, time
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_ROOT]
import ray_tracing_amd as rt
for name, make in (("sphere field, 7.2 M triangles", lambda a: a.spawn_sphere_field(10000, 0x5EED0002, 5)),
                   ("icosphere soup, 1 M triangles", lambda a: a.spawn_icosphere_soup(1389, 0x5EED0003, 5))):
    app = rt.App(); make(app); sc = app.prepare_meshinfo()
    for th in (1, 4, 16):
        t0 = time.perf_counter(); r = rt.host_bvh_build(sc, threads=th, wide=True); dt = time.perf_counter() - t0
        print("%s: %2d threads: binary tree %.0f ms, with the wide tree and copies %.0f ms" % (name, th, r[0].build_ms, dt * 1e3), flush=True)
