"""GPU parity of the live path (ray_tracing.wgsl:173-211) through the C ABI vs the CPU oracle.
Bar: primitive ids equal on >= 99.9 % of pixels and radiance RMSE <= 1e-3;
on these scenes the kernels are expected to be bit-exact, which is asserted where it holds."""
import numpy as np
import pytest

import oracle_py as orc
import ray_tracing_amd as rt
from helpers import (bitwise_equal, default_app, gpu_render, oracle_scene, oracle_view, parity, soup_app)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("flags", [0, rt.FLAG_SHADOW, rt.FLAG_BRUTE_FORCE, rt.FLAG_BRUTE_FORCE | rt.FLAG_SHADOW,
                                   rt.FLAG_NO_MESH_CULL, rt.FLAG_COUNTERS | rt.FLAG_SHADOW,
                                   rt.FLAG_STATIC_GRID, rt.FLAG_STATIC_GRID | rt.FLAG_SHADOW,
                                   rt.FLAG_WAVE_TRAVERSAL, rt.FLAG_WAVE_TRAVERSAL | rt.FLAG_SHADOW,
                                   rt.FLAG_WAVE_TRAVERSAL | rt.FLAG_SHADOW | rt.FLAG_COUNTERS])
def test_default_scene_640x360(flags):
    app = default_app(640, 360)
    scene = app.prepare_meshinfo()
    oflags = (orc.FLAG_SHADOW if flags & rt.FLAG_SHADOW else 0) | (orc.FLAG_NO_MESH_CULL if flags & rt.FLAG_NO_MESH_CULL else 0)
    rgba_o, prim_o, t_o, rays_o = orc.render_primary(oracle_scene(scene), oracle_view(app), oflags)
    ctx, rgba_g, prim_g, t_g = gpu_render(app, scene, flags)
    agree, rmse = parity(rgba_g, prim_g, rgba_o, prim_o)
    assert agree >= 0.999 and rmse <= 1e-3, (agree, rmse)
    assert bitwise_equal(prim_g, prim_o)
    assert bitwise_equal(t_g, t_o)
    assert bitwise_equal(rgba_g, rgba_o)
    c = ctx.counters()
    assert c.rays_primary == 640 * 360
    assert c.rays_primary + c.rays_shadow == rays_o
    if flags & rt.FLAG_COUNTERS:
        assert c.node_visits > 0 and c.tri_tests > 0
    ctx.close()


def test_soup_bvh_matches_oracle_and_brute_force():
    app = soup_app(40, 0x5EED0003, 320, 180)
    scene = app.prepare_meshinfo()
    rgba_o, prim_o, t_o, _ = orc.render_primary(oracle_scene(scene), oracle_view(app), orc.FLAG_SHADOW)
    ctx, rgba_g, prim_g, t_g = gpu_render(app, scene, rt.FLAG_SHADOW)
    agree, rmse = parity(rgba_g, prim_g, rgba_o, prim_o)
    assert agree >= 0.999 and rmse <= 1e-3, (agree, rmse)
    assert bitwise_equal(prim_g, prim_o) and bitwise_equal(t_g, t_o) and bitwise_equal(rgba_g, rgba_o)
    for f in (rt.FLAG_BRUTE_FORCE, rt.FLAG_WAVE_TRAVERSAL, rt.FLAG_STATIC_GRID):
        _, rgba_b, prim_b, t_b = gpu_render(app, scene, rt.FLAG_SHADOW | f, ctx=ctx)
        assert bitwise_equal(prim_b, prim_o) and bitwise_equal(t_b, t_o) and bitwise_equal(rgba_b, rgba_o)
    ctx.close()
