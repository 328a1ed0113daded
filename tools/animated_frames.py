"""Experiment script (GPU): the reference's own per-frame workload -- the default scene with its rotating cube
(src/main.rs:119-126), one RayTracingPassNode::run per frame through the host mirror: prepare_meshinfo, find what moved,
rt_update_vertex_range, rt_render.  Wall time per frame, and the same on the sphere field with one sphere tagged. Not a test."""
import os, sys, time
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_ROOT]
import numpy as np, ray_tracing_amd as rt

def run(name, app, frames=200, entity=None):
    all_ents = app.entities() if entity == "all" else None
    ent = app.entities()[entity] if isinstance(entity, int) else None
    app.camera_set_target(1920, 1080)
    app.finish(0)
    assert app.update(flags=rt.FLAG_SHADOW) == rt.RT_OK           # first frame: upload + build
    ctx = app.ctx()
    ctx.read_prim_ids()
    t_static = time.perf_counter()
    for _ in range(frames): app.update(flags=rt.FLAG_SHADOW)      # nothing moves: no upload at all
    ctx.read_hit_t(); t_static = (time.perf_counter() - t_static) / frames
    t0 = time.perf_counter()
    for f in range(frames):
        if entity == "all":
            a = 0.5 * (f + 1) / 60.0
            for k, e in enumerate(all_ents): app.set_transform(k, e["scale"], (0.0, float(np.sin(a)), 0.0, float(np.cos(a))), e["translation"])
        elif entity is None: app.rotate_tagged(1.0 / 60.0)
        else:
            a = 0.5 * (f + 1) / 60.0
            app.set_transform(entity, ent["scale"], (0.0, float(np.sin(a)), 0.0, float(np.cos(a))), ent["translation"])
        app.update(flags=rt.FLAG_SHADOW)
    ctx.read_hit_t(); dt = (time.perf_counter() - t0) / frames
    print("%s: %.3f ms per frame with the tagged mesh rotating (static scene: %.3f ms), updates (range, full) = %s" % (
        name, dt * 1e3, t_static * 1e3, app.update_counts()), flush=True)
    app.close()

app = rt.App(); app.spawn_default_scene(); run("default scene, 2,894 triangles", app)
app = rt.App(); app.spawn_sphere_field(2000, 0x5EED0002, 5); app.camera_frame_scene(1.0)
run("sphere field, 1.44 M triangles, sphere 7 spinning", app, frames=50, entity=7)
app = rt.App(); app.spawn_sphere_field(2000, 0x5EED0002, 5); app.camera_frame_scene(1.0)
run("sphere field, 1.44 M triangles, every entity spinning (Python sets 2,001 transforms per frame)", app, frames=20, entity="all")
