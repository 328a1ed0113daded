"""The oracle's tree walk (oracle/rt_oracle.c orc_hit_tree: a scalar restatement of the product's traversal, no
reference counterpart -- src/bvh.rs:1 is empty) against the oracle's reference-shaped loop (wgsl:150-170), on the
CPU: over the product's binary tree and over its 4-wide quantised tree the walk must find the same hit for every
ray, bit for bit.  This is the CPU half of SURVEY 8d's "cross-checked by the CPU BVH oracle"; the GPU half
(tests/test_gpu_paths.py) compares the walk's node / triangle counts with the kernel's counters.
Parity unpinned by the reference (it has no tests); pinned by the oracle's own known-answer tests."""
import numpy as np
import pytest

import oracle_py as orc
import ray_tracing_amd as rt
from helpers import default_app, soup_app, field_app, oracle_scene, oracle_view


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def _trees(scene):
    info, nodes, recs, wn, ws, need = rt.host_bvh_build(scene, threads=2, wide=True)
    return orc.Tree(nodes, recs), orc.Tree(nodes, recs, wn)


def _apps():
    yield "default", default_app(160, 90)
    yield "soup", soup_app(12, 0x5EED0003, 96, 64, subdivisions=2)
    yield "field", field_app(40, 0x5EED0002, 96, 54, subdivisions=1)


@pytest.mark.parametrize("name", ["default", "soup", "field"])
@pytest.mark.parametrize("flags", [0, orc.FLAG_SHADOW, orc.FLAG_SHADOW | orc.FLAG_NO_MESH_CULL])
def test_tree_walk_equals_the_reference_shaped_loop(name, flags):
    app = dict(_apps())[name]
    scene = app.prepare_meshinfo()
    osc, view = oracle_scene(scene), oracle_view(app)
    rgba_o, prim_o, t_o, rays_o = orc.render_primary(osc, view, flags, threads=4)
    assert (prim_o != orc.MISS_PRIM).any()
    visits = []
    for tree in _trees(scene):
        rgba, prim, t, rays, (nv, nt, nc) = orc.render_primary_tree(osc, tree, view, flags, threads=4)
        assert rays == rays_o
        assert (prim == prim_o).all() and (_bits(t) == _bits(t_o)).all() and (_bits(rgba) == _bits(rgba_o)).all()
        assert nv > 0 and nt > 0 and (nc > 0) == (not flags & orc.FLAG_NO_MESH_CULL)
        visits.append(nv)
    assert visits[1] < visits[0]          # four children per step: fewer node visits on the wide tree


def test_tree_walk_counts_do_not_depend_on_threads_or_on_the_pixel_list():
    app = default_app(96, 54)
    scene = app.prepare_meshinfo()
    osc, view = oracle_scene(scene), oracle_view(app)
    _, wide = _trees(scene)
    rgba1, prim1, t1, rays1, cnt1 = orc.render_primary_tree(osc, wide, view, orc.FLAG_SHADOW, threads=1)
    rgba4, prim4, t4, rays4, cnt4 = orc.render_primary_tree(osc, wide, view, orc.FLAG_SHADOW, threads=4)
    assert cnt1 == cnt4 and rays1 == rays4 and (prim1 == prim4).all()
    pix = np.arange(96 * 54, dtype=np.uint32)[::-1]
    rgba_l, prim_l, t_l, rays_l, cnt_l = orc.render_primary_tree(osc, wide, view, orc.FLAG_SHADOW, threads=3, pixel_list=pix)
    assert cnt_l == cnt1 and (prim_l == prim1.ravel()[pix]).all()


def test_single_rays_through_the_tree():
    app = default_app(64, 36)
    scene = app.prepare_meshinfo()
    osc = oracle_scene(scene)
    binary, wide = _trees(scene)
    rng = np.random.default_rng(11)
    for i in range(200):
        o = rng.uniform(-6, 6, 3).astype(np.float32)
        if i % 5 == 0:          # axis-parallel rays: infinite reciprocals (the shadow rays are (0,1,0))
            d = np.zeros(3, np.float32)
            d[rng.integers(3)] = rng.choice([-1.0, 1.0])
        else:
            d = (rng.uniform(-1, 1, 3).astype(np.float32) - o * np.float32(0.2)).astype(np.float32)
            d = (d / np.linalg.norm(d)).astype(np.float32)
        ref = orc.hit_triangles(osc, o, d)
        for tree in (binary, wide):
            h, cnt = orc.hit_tree(osc, tree, o, d)
            assert h["hit"] == ref["hit"] and h["prim"] == ref["prim"], (i, o, d)
            if ref["hit"]:
                assert np.float32(h["t"]) == np.float32(ref["t"]) and (h["normal"] == ref["normal"]).all()
                assert h["material"] == ref["material"]
            a, _ = orc.hit_tree(osc, tree, o, d, any_hit=True)
            assert a["hit"] == ref["hit"]


def test_tree_walk_edge_cases():
    empty = rt.Scene(np.zeros(0, rt.TRIANGLE), np.zeros(0, rt.MESH_INFO), np.zeros(0, rt.VERTEX), np.zeros(0, rt.MATERIAL))
    info, nodes, recs, wn, ws, need = rt.host_bvh_build(empty, wide=True)
    osc = oracle_scene(empty)
    for tree in (orc.Tree(nodes, recs), orc.Tree(nodes, recs, wn)):
        h, cnt = orc.hit_tree(osc, tree, (0, 0, 0), (0, 0, 1))
        assert not h["hit"] and h["prim"] == orc.MISS_PRIM and cnt == (0, 0, 0)
    # one triangle: a root with a single child; a ray that starts ON the box's lower planes (distance -0 / +0)
    v = np.zeros(3, rt.VERTEX)
    v["pos"] = [(0, 0, 0), (0, 0, 1), (1, 0, 0)]
    v["norm"] = (0, 1, 0)
    t = np.zeros(1, rt.TRIANGLE)
    t["indices"] = (0, 1, 2)
    m = np.zeros(1, rt.MESH_INFO)
    m["tri_count"] = 1
    m["aabb_min"], m["aabb_max"] = (0, 0, 0), (1, 0, 1)
    one = rt.Scene(t, m, v, np.ones(1, rt.MATERIAL))
    info, nodes, recs, wn, ws, need = rt.host_bvh_build(one, wide=True)
    osc = oracle_scene(one)
    for o, d in (((0.25, 1, 0.25), (0, -1, 0)), ((0.25, 0, 0.25), (0, -1, 0)), ((0, 2, 0), (0, -1, 0)), ((0.25, -1, 0.25), (0, 1, 0))):
        ref = orc.hit_triangles(osc, o, d)
        for tree in (orc.Tree(nodes, recs), orc.Tree(nodes, recs, wn)):
            h, cnt = orc.hit_tree(osc, tree, o, d)
            assert (h["hit"], h["prim"]) == (ref["hit"], ref["prim"]), (o, d)
            assert cnt[0] >= 1


def test_path_tracing_through_the_tree_equals_the_reference_shaped_loop():
    app = soup_app(12, 0x5EED0003, 64, 48, subdivisions=2)
    scene = app.prepare_meshinfo()
    osc, view = oracle_scene(scene), oracle_view(app)
    acc_o, segs_o = orc.render_path(osc, view, spp=3, max_segments=4, seed=7, stream=1, threads=4)
    for tree in _trees(scene):
        acc, segs, cnt = orc.render_path_tree(osc, tree, view, spp=3, max_segments=4, seed=7, stream=1, threads=4)
        assert segs == segs_o and (_bits(acc) == _bits(acc_o)).all() and cnt[0] > 0


def test_cropped_path_frames_hold_the_whole_frames_values():
    """orc_render_path_tree_crop (what the full-size configs[3] / configs[4] GPU tests check against): the pixels of a
    rectangle get exactly the values they have in the whole frame, with and without a viewport, at the frame's edges too."""
    app = soup_app(12, 0x5EED0003, 64, 48, subdivisions=2)
    scene = app.prepare_meshinfo()
    osc = oracle_scene(scene)
    _, wide = _trees(scene)
    m, p = app.camera_view()
    for viewport in (None, (5, 7, 50, 30)):
        view = orc.make_view(m, p, 64, 48, viewport=viewport)
        whole, _, _ = orc.render_path_tree(osc, wide, view, spp=3, max_segments=4, seed=7, sample_first=2, threads=4)
        for crop in ((0, 0, 64, 48), (0, 0, 9, 5), (20, 11, 17, 23), (64 - 6, 48 - 4, 6, 4), (3, 40, 61, 8)):
            x, y, w, h = crop
            part = orc.render_path_tree_crop(osc, wide, view, crop, spp=3, max_segments=4, seed=7, sample_first=2, threads=3)
            assert part.shape == (h, w, 4) and (_bits(part) == _bits(whole[y:y + h, x:x + w])).all()
