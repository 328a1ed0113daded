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


def _analytic_case():
    app = rt.App()
    app.spawn_sphere_field(30, 0x5EED0002, 1)
    app.camera_frame_scene()
    app.camera_set_target(160, 90)
    app.prepare_meshinfo()
    return app, rt.analytic_sphere_field(30, 0x5EED0002)


@pytest.mark.parametrize("name", ["default", "soup", "analytic"])
def test_counters_equal_the_oracle_tree_walk(name):
    """the node visits / triangle tests behind roofline.achieved are COUNTED by the kernel and cross-checked
    by the CPU oracle walking the same tree in the same order (oracle/rt_oracle.c orc_hit_tree): equal to the unit,
    on the wide tree and on the binary tree, from the persistent kernel and from the static grid."""
    if name == "default":
        app = default_app(320, 180)
        scene = app.prepare_meshinfo()
    elif name == "soup":
        app = soup_app(40, 0x5EED0003, 160, 90, subdivisions=3)
        scene = app.prepare_meshinfo()
    else:
        app, scene = _analytic_case()
    osc = orc.Scene(scene.triangles, scene.meshes, scene.vertices, scene.materials,
                    getattr(scene, "spheres", None), getattr(scene, "sphere_materials", None))
    view = oracle_view(app)
    ctx = rt.Context(0)
    ctx.upload_scene(scene)
    nodes, recs = ctx.read_bvh()                 # the tree the device walks, read back through the C ABI
    wide = ctx.read_bvh_wide()
    assert len(wide) > 0
    for tree_flag, tree in ((0, orc.Tree(nodes, recs, wide)), (rt.FLAG_BINARY_TREE, orc.Tree(nodes, recs))):
        for extra, oflags in ((rt.FLAG_SHADOW, orc.FLAG_SHADOW), (rt.FLAG_SHADOW | rt.FLAG_NO_MESH_CULL, orc.FLAG_SHADOW | orc.FLAG_NO_MESH_CULL), (0, 0)):
            rgba_o, prim_o, t_o, rays_o, (nv, nt, nc) = orc.render_primary_tree(osc, tree, view, oflags)
            for kernel in (0, rt.FLAG_STATIC_GRID):
                _, rgba_g, prim_g, t_g = gpu_render(app, scene, extra | tree_flag | kernel | rt.FLAG_COUNTERS, ctx=ctx)
                c = ctx.counters()
                assert bitwise_equal(prim_g, prim_o) and bitwise_equal(t_g, t_o) and bitwise_equal(rgba_g, rgba_o)
                assert c.rays_primary + c.rays_shadow == rays_o
                assert (c.node_visits, c.tri_tests, c.mesh_aabb_tests) == (nv, nt, nc), (name, tree_flag, extra, kernel)
    ctx.close()


def test_two_contexts_on_one_gpu_render_concurrently():
    """Two contexts = two streams and two sets of frame buffers: frames submitted alternately overlap on the device (the
    way to keep two frames in flight); each context must still deliver its own frame, bit for bit."""
    app = default_app(320, 180)
    scene = app.prepare_meshinfo()
    m, p = app.camera_view()
    app2 = default_app(320, 180)
    app2.rotate_tagged(0.4)                        # the second context renders a different pose of the cube
    scene2 = app2.prepare_meshinfo()
    ref = [orc.render_primary(oracle_scene(s), oracle_view(app), orc.FLAG_SHADOW) for s in (scene, scene2)]
    with rt.Context(0) as a, rt.Context(0) as b:
        for ctx, s in ((a, scene), (b, scene2)):
            ctx.upload_scene(s)
            ctx.set_view(m, p, 320, 180)
        for _ in range(20):                        # no synchronisation between the submissions
            assert a.render(flags=rt.FLAG_SHADOW) == rt.RT_OK
            assert b.render(flags=rt.FLAG_SHADOW) == rt.RT_OK
        for ctx, (rgba_o, prim_o, t_o, _) in ((a, ref[0]), (b, ref[1])):
            assert bitwise_equal(ctx.read_prim_ids(), prim_o) and bitwise_equal(ctx.read_radiance(), rgba_o)
        assert not bitwise_equal(a.read_prim_ids(), b.read_prim_ids())


@pytest.mark.parametrize("size", [(4096, 1), (1, 2048), (3, 5), (1, 1), (2047, 3)])
def test_extreme_frame_shapes(size):
    """Frames far from tile multiples: one row, one column, a handful of pixels -- every kernel variant, shadow rays and a
    path-traced sample, against the oracle."""
    W, H = size
    app = default_app(W, H)
    scene = app.prepare_meshinfo()
    osc, view = oracle_scene(scene), oracle_view(app)
    rgba_o, prim_o, t_o, rays_o = orc.render_primary(osc, view, orc.FLAG_SHADOW)
    acc_o, segs_o = orc.render_path(osc, view, 5, 4, seed=2)
    ctx = None
    for extra in (0, rt.FLAG_STATIC_GRID, rt.FLAG_WAVE_TRAVERSAL, rt.FLAG_BINARY_TREE):
        ctx, rgba_g, prim_g, t_g = gpu_render(app, scene, rt.FLAG_SHADOW | extra, ctx=ctx) if ctx else gpu_render(app, scene, rt.FLAG_SHADOW | extra)
        assert bitwise_equal(prim_g, prim_o) and bitwise_equal(t_g, t_o) and bitwise_equal(rgba_g, rgba_o), (size, extra)
        assert ctx.counters().rays_primary == W * H
    assert ctx.render(flags=rt.FLAG_PATH_TRACE, spp=5, max_segments=4, seed=2) == rt.RT_OK
    assert bitwise_equal(ctx.read_accumulation(), acc_o)
    ctx.close()
