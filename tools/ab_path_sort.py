"""Experiment script (GPU): bounce queue regrouped by ray direction inside the bounce kernel (RT_TUNE_PATH_SORT / _KEY)
against the plain k_path_bounce: ms per frame, wave trips, lanes alive per node trip; running sums must stay bit-identical.
Not a test."""
import os, sys
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_ROOT]
import numpy as np, ray_tracing_amd as rt

SETTINGS = [(0, 0)] + [(m, k) for k in (0, 1) for m in (1, 2, 4, 6)]
SCENES = (("soup1M_4K_4seg", 1389, 0x5EED0003, 3840, 2160, 4, 1), ("soup10M_1080p_8seg", 13889, 0x5EED0005, 1920, 1080, 8, 1),
          ("soup1M_1080p_8seg_4spp", 1389, 0x5EED0003, 1920, 1080, 8, 4))
only = os.environ.get("AB_SCENES")
for name, n, seed, W, H, segs, spp in SCENES:
    if only and name not in only.split(","):
        continue
    app = rt.App(); app.spawn_icosphere_soup(n, seed, 5); app.camera_frame_scene(1.0); app.camera_set_target(W, H)
    sc = app.prepare_meshinfo(); m4, p = app.camera_view()
    base = None
    for m, key in SETTINGS:
        os.environ["RT_TUNE_PATH_SORT"] = str(m); os.environ["RT_TUNE_PATH_SORT_KEY"] = str(key)
        ctx = rt.Context(0); ctx.upload_scene(sc); ctx.set_view(m4, p, W, H)
        ctx.render(flags=rt.FLAG_PATH_TRACE | rt.FLAG_COUNTERS, spp=spp, max_segments=segs, seed=1); c = ctx.counters()
        acc = ctx.read_accumulation()
        if base is None:
            base = acc
        same = acc.tobytes() == base.tobytes()
        for _ in range(2): ctx.render(flags=rt.FLAG_PATH_TRACE, spp=spp, max_segments=segs, seed=1)
        ctx.counters()
        for _ in range(5): ctx.render(flags=rt.FLAG_PATH_TRACE | rt.FLAG_TIME_KERNELS, spp=spp, max_segments=segs, seed=1)
        k = ctx.counters(); rays = k.rays_primary + k.rays_bounce
        print("%-24s sort %d key %d  ms/frame %7.3f  Mrays/s %6.0f  alive/node trip %.3f  wave trips node %d tri %d  identical %s" % (
            name, m, key, k.kernel_ms / 5, rays / (k.kernel_ms / 5) / 1e3, c.wave_alive_lanes / max(1, 64 * c.wave_node_iters),
            c.wave_node_iters, c.wave_tri_iters, same), flush=True)
        ctx.close()
