"""Experiment script (GPU): BASELINE.json configs[4] resolution (8K) end to end on one GPU: index arithmetic,
buffer sizes, path queues; sampled parity against the oracle. Not a test (too big for the suite)."""
import os, sys, time
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_ROOT, os.path.join(_ROOT, "oracle")]
import numpy as np, ray_tracing_amd as rt, oracle_py as orc
W, H = 7680, 4320
app = rt.App(); app.spawn_icosphere_soup(13889, 0x5EED0005, 5); app.camera_frame_scene(1.0); app.camera_set_target(W, H)
sc = app.prepare_meshinfo(); m, p = app.camera_view()
ctx = rt.Context(0); t0 = time.perf_counter(); ctx.upload_scene(sc); print("upload+build %.1f s" % (time.perf_counter() - t0), flush=True)
ctx.set_view(m, p, W, H)
ctx.render(flags=rt.FLAG_SHADOW); ctx.counters()
for _ in range(3): ctx.render(flags=rt.FLAG_SHADOW | rt.FLAG_TIME_KERNELS)
k = ctx.counters(); rays = k.rays_primary + k.rays_shadow
print("8K primary+shadow, 10 M triangles: %.2f ms/frame, %.0f Mrays/s" % (k.kernel_ms / 3, rays / (k.kernel_ms / 3) / 1e3), flush=True)
prim = ctx.read_prim_ids().ravel(); rgba = ctx.read_radiance().reshape(-1, 4)
rng = np.random.default_rng(1); pix = rng.choice(W * H, 20000, replace=False).astype(np.uint32)
osc = orc.Scene(sc.triangles, sc.meshes, sc.vertices, sc.materials); view = orc.make_view(m, p, W, H)
rgba_o, prim_o, _, _ = orc.render_primary(osc, view, orc.FLAG_SHADOW, threads=16, pixel_list=pix)
print("sampled parity: prim agreement %.5f, radiance identical %s" % (float((prim[pix] == prim_o).mean()), bool((rgba[pix].view(np.uint32) == rgba_o.view(np.uint32)).all())), flush=True)
ctx.render(flags=rt.FLAG_PATH_TRACE, spp=1, max_segments=8, seed=1); ctx.counters()
ctx.render(flags=rt.FLAG_PATH_TRACE | rt.FLAG_TIME_KERNELS, spp=1, max_segments=8, seed=1)
k = ctx.counters(); rays = k.rays_primary + k.rays_bounce
print("8K path tracing 1 spp <= 8 segments: %.1f ms, %.0f Mrays/s, %d segments" % (k.kernel_ms, rays / k.kernel_ms / 1e3, rays), flush=True)
acc = ctx.read_accumulation().reshape(-1, 4)
print("alpha == 1 everywhere:", bool((acc[:, 3] == 1.0).all()), "finite:", bool(np.isfinite(acc).all()), flush=True)
