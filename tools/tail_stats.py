"""Experiment script (GPU): share of the node-loop trips a wave spends with few unfinished rays. Not a test."""
import os, sys
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_ROOT]
os.environ["RT_DEBUG_TAIL"] = "1"; os.environ["RT_TUNE_TRAVERSE"] = "0"
import ray_tracing_amd as rt
for kind in ("field", "soup", "default"):
    app = rt.App()
    if kind == "field": app.spawn_sphere_field(10000, 0x5EED0002, 5); app.camera_frame_scene(1.0)
    elif kind == "soup": app.spawn_icosphere_soup(1389, 0x5EED0003, 5); app.camera_frame_scene(1.0)
    else: app.spawn_default_scene()
    app.camera_set_target(1920, 1080)
    sc = app.prepare_meshinfo(); m, p = app.camera_view()
    ctx = rt.Context(0); ctx.upload_scene(sc); ctx.set_view(m, p, 1920, 1080)
    print(kind, flush=True)
    ctx.render(flags=rt.FLAG_SHADOW | rt.FLAG_COUNTERS); ctx.counters(); ctx.close()
