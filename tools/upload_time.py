"""Experiment script (GPU): wall time of rt_upload_scene (build + tables + copies) for the bench scene. Not a test."""
import os, sys, time
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_ROOT]
import ray_tracing_amd as rt
app = rt.App(); app.spawn_sphere_field(10000, 0x5EED0002, 5)
t0 = time.perf_counter(); sc = app.prepare_meshinfo(); print("prepare_meshinfo (host, 7.2 M triangles): %.0f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
ctx = rt.Context(0)
for rep in range(2):
    t0 = time.perf_counter(); ctx.upload_scene(sc); dt = time.perf_counter() - t0
    i = ctx.bvh_info()
    print("rt_upload_scene: %.0f ms wall (binary tree build %.0f ms)" % (dt * 1e3, i.build_ms), flush=True)
ctx.close()
