"""Analytic spheres (EXTENSION, SURVEY 8d config 2 variant ii).  The reference has no ray/sphere test
(its live spheres are tessellated, src/main.rs:85; the Sphere type of the dead src/scene.rs:26-28 only
holds centre + radius), so the test is build-defined and PARITY IS UNPINNED by the reference: the oracle
is pinned here by hand-worked known answers, the GPU paths against the oracle."""
import numpy as np
import pytest

import oracle_py as orc
import ray_tracing_amd as rt
from helpers import bitwise_equal


def _one_sphere(center=(0.0, 0.0, 0.0), radius=1.0):
    tris = np.zeros(0, orc.TRIANGLE); meshes = np.zeros(0, orc.MESH_INFO); verts = np.zeros(0, orc.VERTEX)
    mats = np.zeros(1, orc.MATERIAL); mats["color"][0] = (0.2, 0.4, 0.8, 1.0)
    sp = np.zeros(1, orc.SPHERE); sp["center"][0] = center; sp["radius"][0] = radius
    return orc.Scene(tris, meshes, verts, mats, sp, np.zeros(1, np.uint32))


def test_kat_sphere_entry_hit_normal_and_shade():
    s = _one_sphere()
    h = orc.hit_triangles(s, [0.0, 3.0, 0.0], [0.0, -1.0, 0.0])
    # oc=(0,3,0) a=1 hb=-3 cc=8 disc=1 t=(3-1)/1=2, point (0,1,0), normal (0,1,0)
    assert h["hit"] and h["t"] == 2.0 and h["prim"] == 0 and h["material"] == 0
    assert np.array_equal(h["point"], [0, 1, 0]) and np.array_equal(h["normal"], [0, 1, 0])
    assert np.allclose(orc.shade(s, h), [0.2, 0.4, 0.8, 1.0])


def test_kat_sphere_inside_origin_and_behind_miss():
    s = _one_sphere()
    assert not orc.hit_triangles(s, [0.0, 0.0, 0.0], [0.0, 1.0, 0.0])["hit"]    # origin inside: entry t < 0
    assert not orc.hit_triangles(s, [0.0, 3.0, 0.0], [0.0, 1.0, 0.0])["hit"]    # sphere behind the ray
    assert not orc.hit_triangles(s, [2.0, 3.0, 0.0], [0.0, -1.0, 0.0])["hit"]   # passes beside it (disc < 0)
    g = orc.hit_triangles(s, [1.0, 3.0, 0.0], [0.0, -1.0, 0.0])                 # grazing: disc == 0 counts
    assert g["hit"] and g["t"] == 3.0


def test_kat_sphere_unnormalised_direction():
    s = _one_sphere()
    h = orc.hit_triangles(s, [0.0, 3.0, 0.0], [0.0, -2.0, 0.0])     # a=4 hb=-6 cc=8 disc=4 t=(6-2)/4=1
    assert h["hit"] and h["t"] == 1.0 and np.array_equal(h["point"], [0, 1, 0])


def test_kat_triangle_wins_a_tie_and_prim_id_offset():
    # a triangle in the plane y=1 and a unit sphere touching it from below: both at t=2; the triangle (earlier in
    # loop order) keeps the tie, as the strict compare of wgsl:164 does between triangles
    tris = np.zeros(1, orc.TRIANGLE); tris["indices"][0] = (0, 1, 2)
    verts = np.zeros(3, orc.VERTEX)
    verts["pos"][:] = [(-1, 1, -1), (-1, 1, 1), (1, 1, -1)]; verts["norm"][:] = (0, 1, 0)
    meshes = np.zeros(1, orc.MESH_INFO)
    meshes["first_tri"], meshes["tri_count"], meshes["material"] = 0, 1, 0
    meshes["aabb_min"][0], meshes["aabb_max"][0] = (-1, 0.9, -1), (1, 1.1, 1)
    mats = np.zeros(2, orc.MATERIAL); mats["color"][:] = 1.0
    sp = np.zeros(1, orc.SPHERE); sp["radius"][0] = 1.0
    s = orc.Scene(tris, meshes, verts, mats, sp, np.ones(1, np.uint32))
    h = orc.hit_triangles(s, [-0.5, 3.0, -0.5], [0.0, -1.0, 0.0])
    assert h["hit"] and h["prim"] == 0 and h["material"] == 0
    h = orc.hit_triangles(s, [0.5, 3.0, 0.5], [0.0, -1.0, 0.0])      # beside the triangle: the sphere, prim = n_tri + 0
    assert h["hit"] and h["prim"] == 1 and h["material"] == 1


def test_analytic_field_matches_the_tessellated_field():
    n, seed = 40, 0x5EED0002
    sc = rt.analytic_sphere_field(n, seed)
    app = rt.App(); app.spawn_sphere_field(n, seed, 2)
    tess = app.prepare_meshinfo()
    assert sc.spheres.size == n and sc.materials.size == tess.materials.size
    assert np.array_equal(sc.materials["color"], tess.materials["color"])
    for k in range(n):      # mesh k+1 of the tessellated field is sphere k: same centre, vertices at the radius
        m = tess.meshes[k + 1]
        tri = tess.triangles["indices"][m["first_tri"]:m["first_tri"] + m["tri_count"]]
        p = tess.vertices["pos"][np.unique(tri)]
        d = np.linalg.norm(p.astype(np.float64) - sc.spheres["center"][k].astype(np.float64), axis=1)
        assert np.allclose(d, sc.spheres["radius"][k], rtol=1e-5)


def test_host_bvh_boxes_contain_the_spheres():
    sc = rt.analytic_sphere_field(300, 7)
    ctx_free = rt.lib()      # host-only: no GPU touched
    import ctypes as C
    h, info = C.c_void_p(), rt.BvhInfo()
    # rt_host_bvh_build has no sphere arguments: the builder's sphere path is exercised through the GPU upload
    # (test below); here only that the scene's triangle part still builds
    rc = ctx_free.rt_host_bvh_build(sc.triangles.ctypes.data, sc.triangles.size, sc.meshes.ctypes.data, sc.meshes.size,
                                    sc.vertices.ctypes.data, sc.vertices.size, 1, C.byref(h), C.byref(info))
    assert rc == rt.RT_OK and info.n_triangles == sc.triangles.size
    ctx_free.rt_host_bvh_free(h)


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [0, rt.FLAG_STATIC_GRID, rt.FLAG_WAVE_TRAVERSAL, rt.FLAG_BRUTE_FORCE])
@pytest.mark.parametrize("shadow", [0, rt.FLAG_SHADOW])
def test_gpu_analytic_spheres_match_the_oracle(extra, shadow):
    n, seed, W, H = 400, 0x5EED0002, 320, 180
    sc = rt.analytic_sphere_field(n, seed)
    app = rt.App(); app.spawn_sphere_field(n, seed, 1); app.camera_frame_scene(1.0); app.camera_set_target(W, H)
    m, p = app.camera_view()
    osc = orc.Scene(sc.triangles, sc.meshes, sc.vertices, sc.materials, sc.spheres, sc.sphere_materials)
    view = orc.make_view(m, p, W, H)
    rgba_o, prim_o, t_o, rays_o = orc.render_primary(osc, view, orc.FLAG_SHADOW if shadow else 0)
    with rt.Context(0) as ctx:
        ctx.upload_scene(sc)
        ctx.set_view(m, p, W, H)
        assert ctx.render(flags=shadow | extra | rt.FLAG_COUNTERS) == rt.RT_OK
        prim, t, rgba = ctx.read_prim_ids(), ctx.read_hit_t(), ctx.read_radiance()
        c = ctx.counters()
        assert (prim >= sc.triangles.size).sum() > 1000                 # spheres are actually visible
        assert float((prim == prim_o).mean()) >= 0.999                  # the stated gate
        assert bitwise_equal(prim, prim_o) and bitwise_equal(t, t_o) and bitwise_equal(rgba, rgba_o)   # in fact bit-exact
        assert c.rays_primary + c.rays_shadow == rays_o
        info = ctx.bvh_info()
        assert info.n_triangles == sc.triangles.size + n                # leaf records: triangles + spheres
        nodes, recs = ctx.read_bvh()
        sph = recs[recs["mesh"] == 0xFFFFFFFF]
        assert sph.size == n and np.array_equal(np.sort(sph["tri"]), np.arange(n) + sc.triangles.size)


@pytest.mark.gpu
def test_gpu_analytic_spheres_path_tracing_matches_the_oracle():
    n, seed, W, H = 150, 11, 128, 72
    sc = rt.analytic_sphere_field(n, seed)
    app = rt.App(); app.spawn_sphere_field(n, seed, 1); app.camera_frame_scene(1.0); app.camera_set_target(W, H)
    m, p = app.camera_view()
    osc = orc.Scene(sc.triangles, sc.meshes, sc.vertices, sc.materials, sc.spheres, sc.sphere_materials)
    view = orc.make_view(m, p, W, H)
    acc_o, segs_o = orc.render_path(osc, view, 2, 4, seed=5)
    with rt.Context(0) as ctx:
        ctx.upload_scene(sc)
        ctx.set_view(m, p, W, H)
        assert ctx.render(flags=rt.FLAG_PATH_TRACE, spp=2, max_segments=4, seed=5) == rt.RT_OK
        rad_g, rad_o = ctx.read_radiance(), orc.resolve(acc_o, 2)
        rmse = float(np.sqrt(np.mean((rad_g.astype(np.float64) - rad_o.astype(np.float64)) ** 2)))
        assert rmse <= 1e-3, rmse
        assert bitwise_equal(ctx.read_accumulation(), acc_o)
        c = ctx.counters()
        assert c.rays_primary + c.rays_bounce == segs_o


def test_scene_rs_default_spheres_known_answers():
    """src/scene.rs:22-52 (dead code in the reference): its three analytic spheres through the oracle."""
    sc = rt.scene_rs_default()
    osc = orc.Scene(sc.triangles, sc.meshes, sc.vertices, sc.materials, sc.spheres, sc.sphere_materials)
    h = orc.hit_triangles(osc, [0.0, 0.0, 5.0], [0.0, 0.0, -1.0])      # the unit sphere at the origin, entered at z = 1
    assert h["hit"] and h["t"] == 4.0 and h["prim"] == 0 and h["material"] == 0 and np.array_equal(h["normal"], [0, 0, 1])
    h = orc.hit_triangles(osc, [2.0, 5.0, -1.0], [0.0, -1.0, 0.0])     # the second sphere from above
    assert h["hit"] and h["t"] == 4.0 and h["prim"] == 1
    h = orc.hit_triangles(osc, [10.0, 5.0, 10.0], [0.0, -1.0, 0.0])    # the radius-100 ground sphere
    assert h["hit"] and h["prim"] == 2 and abs(h["point"][1] + 2.0) < 0.02
    assert not orc.hit_triangles(osc, [0.0, 5.0, 5.0], [0.0, 1.0, 0.0])["hit"]


@pytest.mark.gpu
def test_gpu_scene_rs_default_matches_the_oracle():
    sc = rt.scene_rs_default()
    osc = orc.Scene(sc.triangles, sc.meshes, sc.vertices, sc.materials, sc.spheres, sc.sphere_materials)
    app = rt.App(); app.spawn_default_scene(); app.camera_set_target(320, 180)     # the camera of src/main.rs:59-62
    m, p = app.camera_view()
    view = orc.make_view(m, p, 320, 180)
    rgba_o, prim_o, t_o, rays_o = orc.render_primary(osc, view, orc.FLAG_SHADOW)
    acc_o, segs_o = orc.render_path(osc, view, 4, 4, seed=3)
    with rt.Context(0) as ctx:
        ctx.upload_scene(sc)
        ctx.set_view(m, p, 320, 180)
        assert ctx.render(flags=rt.FLAG_SHADOW) == rt.RT_OK
        prim = ctx.read_prim_ids()
        assert set(np.unique(prim)) == {0, 1, 2, rt.MISS_PRIM}
        assert bitwise_equal(prim, prim_o) and bitwise_equal(ctx.read_hit_t(), t_o) and bitwise_equal(ctx.read_radiance(), rgba_o)
        assert ctx.render(flags=rt.FLAG_PATH_TRACE, spp=4, max_segments=4, seed=3) == rt.RT_OK
        assert bitwise_equal(ctx.read_accumulation(), acc_o)
