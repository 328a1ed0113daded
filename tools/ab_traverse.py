"""Experiment script (GPU): A/B of the per-lane traversal shapes and the persistent kernel. Not a test."""
import os
import sys
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_ROOT, os.path.join(_ROOT, "oracle")]
import numpy as np, ray_tracing_amd as rt

RES = tuple(int(v) for v in os.environ.get("AB_RES", "1920x1080").split("x"))
KINDS = os.environ.get("AB_KINDS", "field,soup,default").split(",")


def scene(kind):
    app = rt.App()
    if kind == "field": app.spawn_sphere_field(10000, 0x5EED0002, 5); app.camera_frame_scene(1.0)
    elif kind == "soup": app.spawn_icosphere_soup(1389, 0x5EED0003, 5); app.camera_frame_scene(1.0)
    else: app.spawn_default_scene()
    app.camera_set_target(*RES)
    return app

_ENV0 = {k: v for k, v in os.environ.items() if k.startswith("RT_TUNE_")}
variants = []
for extra in sys.argv[1:]:          # name:ENV=VAL,ENV=VAL:flag
    name, envs, fl = extra.split(":")
    variants.append((name, dict(e.split("=") for e in envs.split(",") if e), int(fl, 0)))
ref = {}
for kind in KINDS:
    app = scene(kind); sc = app.prepare_meshinfo(); m, p = app.camera_view()
    for name, env, fl in variants:
        for k in [k for k in os.environ if k.startswith("RT_TUNE_") and k not in _ENV0]: del os.environ[k]
        os.environ.update(_ENV0)            # every variant starts from the environment the script was given
        for k, v in env.items(): os.environ[k] = v
        ctx = rt.Context(0); ctx.upload_scene(sc); ctx.set_view(m, p, *RES)
        f = rt.FLAG_SHADOW | fl
        ctx.render(flags=f | rt.FLAG_COUNTERS); c = ctx.counters()
        for _ in range(3): ctx.render(flags=f)
        ctx.counters()
        for _ in range(20): ctx.render(flags=f | rt.FLAG_TIME_KERNELS)
        k = ctx.counters(); ms = k.kernel_ms / k.kernel_launches; rays = c.rays_primary + c.rays_shadow
        prim = ctx.read_prim_ids(); rgba = ctx.read_radiance()
        if kind not in ref: ref[kind] = (prim.copy(), rgba.copy())
        same = bool((prim == ref[kind][0]).all() and (rgba.view(np.uint32) == ref[kind][1].view(np.uint32)).all())
        print(kind, name, "ms %.4f Mrays/s %.0f" % (ms, rays / ms / 1e3), "identical" if same else "DIFFERENT",
              "nodes/ray %.1f tris/ray %.2f alive %.2f util node %.2f tri %.2f witers/wave %d %d" % (
              c.node_visits / rays, c.tri_tests / rays, c.wave_alive_lanes / max(1, 64 * c.wave_node_iters),
              c.node_visits / max(1, 64 * c.wave_node_iters), c.tri_tests / max(1, 64 * c.wave_tri_iters),
              c.wave_node_iters / (RES[0] * RES[1] // 64 + 1), c.wave_tri_iters / (RES[0] * RES[1] // 64 + 1)), flush=True)
        ctx.close()
