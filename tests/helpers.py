"""Shared test helpers: scenes through the product's host mirror, views for both sides, parity metrics."""
import numpy as np

import oracle_py as orc
import ray_tracing_amd as rt


def default_app(width=640, height=360, viewport=None):
    app = rt.App()
    app.spawn_default_scene()
    app.camera_set_target(width, height, viewport)
    return app


def soup_app(n, seed, width, height, subdivisions=5):
    app = rt.App()
    app.spawn_icosphere_soup(n, seed, subdivisions)
    app.camera_frame_scene()
    app.camera_set_target(width, height)
    return app


def field_app(n, seed, width, height, subdivisions=5):
    app = rt.App()
    app.spawn_sphere_field(n, seed, subdivisions)
    app.camera_frame_scene()
    app.camera_set_target(width, height)
    return app


def oracle_scene(scene):
    return orc.Scene(scene.triangles, scene.meshes, scene.vertices, scene.materials)


def oracle_view(app, viewport=None):
    m, p = app.camera_view()
    return orc.make_view(m, p, app.width, app.height, viewport)


def gpu_render(app, scene, flags=0, viewport=None, ctx=None, **kw):
    """Upload + view + render through the C ABI; returns (ctx, rgba, prim, t)."""
    own = ctx is None
    if own:
        ctx = rt.Context(0)
        ctx.upload_scene(scene)
    m, p = app.camera_view()
    ctx.set_view(m, p, app.width, app.height, viewport)
    rc = ctx.render(flags=flags, **kw)
    assert rc == rt.RT_OK
    return ctx, ctx.read_radiance(), ctx.read_prim_ids(), ctx.read_hit_t()


def parity(rgba_g, prim_g, rgba_o, prim_o):
    """(fraction of pixels with equal primitive id, linear-radiance RMSE over all channels)."""
    agree = float((prim_g == prim_o).mean())
    a, b = rgba_g.astype(np.float64), rgba_o.astype(np.float64)
    both_nan = np.isnan(a) & np.isnan(b)
    d = np.where(both_nan, 0.0, a - b)
    rmse = float(np.sqrt(np.mean(d * d)))
    return agree, rmse


def bitwise_equal(a, b):
    return a.shape == b.shape and a.tobytes() == b.tobytes()
