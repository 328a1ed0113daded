"""Experiment script (GPU): the 4-wide tree on small scenes with a trip-limited debug build. Not a test."""
import os, sys
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_ROOT, os.path.join(_ROOT, "oracle")]
os.environ["RT_TUNE_WIDE"] = "1"
import numpy as np, ray_tracing_amd as rt
for kind, w, h in (("default", 160, 90), ("soup20", 320, 180), ("soup200", 640, 360)):
    app = rt.App()
    if kind == "default": app.spawn_default_scene()
    else: app.spawn_icosphere_soup(int(kind[4:]), 0x5EED0003, 3); app.camera_frame_scene(1.0)
    app.camera_set_target(w, h)
    sc = app.prepare_meshinfo(); m, p = app.camera_view()
    ctx = rt.Context(0); ctx.upload_scene(sc); ctx.set_view(m, p, w, h)
    ctx.render(flags=rt.FLAG_SHADOW | rt.FLAG_BINARY_TREE); c0 = ctx.counters(); prim0 = ctx.read_prim_ids(); rgba0 = ctx.read_radiance()
    print(kind, "binary: rays", c0.rays_primary, c0.rays_shadow, flush=True)
    ctx.render(flags=rt.FLAG_SHADOW); c1 = ctx.counters(); prim1 = ctx.read_prim_ids(); rgba1 = ctx.read_radiance()
    print(kind, "wide: rays", c1.rays_primary, c1.rays_shadow, "dbg cur/sp %x" % c1.rays_bounce, "t/idx %x" % c1.mesh_aabb_tests,
          "identical", bool((prim0 == prim1).all() and (rgba0.view(np.uint32) == rgba1.view(np.uint32)).all()),
          "diff px", int((prim0 != prim1).sum()), flush=True)
    ctx.render(flags=rt.FLAG_SHADOW | rt.FLAG_COUNTERS | rt.FLAG_BINARY_TREE); a = ctx.counters()
    ctx.render(flags=rt.FLAG_SHADOW | rt.FLAG_COUNTERS); b = ctx.counters()
    print(kind, "nodes/ray binary %.1f wide %.1f tris/ray %.2f %.2f" % (a.node_visits / max(1, a.rays_primary + a.rays_shadow),
          b.node_visits / max(1, b.rays_primary + b.rays_shadow), a.tri_tests / max(1, a.rays_primary + a.rays_shadow),
          b.tri_tests / max(1, b.rays_primary + b.rays_shadow)), flush=True)
    ctx.close()
