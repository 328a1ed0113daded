"""Experiment script (GPU): resident blocks per CU of k_stream vs scene size. Not a test."""
import os, sys
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_ROOT]
import numpy as np, ray_tracing_amd as rt
scenes = [("default", None), ("soup10", 10), ("soup50", 50), ("soup200", 200), ("soup600", 600), ("soup1389", 1389)]
for name, n in scenes:
    app = rt.App()
    if n is None: app.spawn_default_scene()
    else: app.spawn_icosphere_soup(n, 0x5EED0003, 5); app.camera_frame_scene(1.0)
    app.camera_set_target(1920, 1080)
    sc = app.prepare_meshinfo(); m, p = app.camera_view()
    row = []
    for b in (2, 3, 4, 5, 6, 8):
        os.environ["RT_TUNE_BLOCKS_PER_CU"] = str(b)
        ctx = rt.Context(0); ctx.upload_scene(sc); ctx.set_view(m, p, 1920, 1080)
        for _ in range(3): ctx.render(flags=rt.FLAG_SHADOW)
        ctx.counters()
        for _ in range(20): ctx.render(flags=rt.FLAG_SHADOW | rt.FLAG_TIME_KERNELS)
        k = ctx.counters(); row.append("%d:%.3f" % (b, k.kernel_ms / k.kernel_launches))
        info = ctx.bvh_info(); ctx.close()
    print(name, "binary MB %.1f walked (wide nodes + records) MB %.1f" % ((info.n_nodes * 64 + info.n_triangles * 48) / 1e6, (info.n_wide_nodes * 64 + info.n_triangles * 48) / 1e6), " ".join(row), flush=True)
