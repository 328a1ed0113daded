"""Experiment script (GPU): two (or argv[1]) contexts (two streams, two sets of frame buffers) on one GPU rendering alternate frames of the
bench workload, against one context rendering the same number of frames back to back. Not a test."""
import os, sys, time
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_ROOT]
import numpy as np, ray_tracing_amd as rt
app = rt.App(); app.spawn_sphere_field(10000, 0x5EED0002, 5); app.camera_frame_scene(1.0); app.camera_set_target(1920, 1080)
sc = app.prepare_meshinfo(); m, p = app.camera_view()
ctxs = []
NCTX = int(sys.argv[1]) if len(sys.argv) > 1 else 2
for k in range(NCTX):
    c = rt.Context(0); c.upload_scene(sc); c.set_view(m, p, 1920, 1080); ctxs.append(c)
ctxs[0].render(flags=rt.FLAG_SHADOW | rt.FLAG_COUNTERS); cnt = ctxs[0].counters(); rays = cnt.rays_primary + cnt.rays_shadow
for c in ctxs:
    for _ in range(5): c.render(flags=rt.FLAG_SHADOW)
    c.synchronize()
N = 200
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(N): ctxs[0].render(flags=rt.FLAG_SHADOW)
    ctxs[0].synchronize(); one = time.perf_counter() - t0
    t0 = time.perf_counter()
    for f in range(N):
        ctxs[f % NCTX].render(flags=rt.FLAG_SHADOW)
    for c in ctxs: c.synchronize()
    two = time.perf_counter() - t0
    print("one context: %.3f ms/frame = %.0f Mrays/s; %d contexts, frames in turn: %.3f ms/frame = %.0f Mrays/s" % (
        one / N * 1e3, rays * N / one / 1e6, NCTX, two / N * 1e3, rays * N / two / 1e6), flush=True)
a, b = ctxs[0].read_prim_ids(), ctxs[1].read_prim_ids()
print("same frame from both:", bool((a == b).all()))
for c in ctxs: c.close()
