"""GPU parity (through the C ABI) of the wider hot-path rows: path tracing + accumulation (A2, A7, A8),
viewport and tile sharding (L1), presentation (f2), the plugin lifecycle (L1/L2/P1), edge cases, golden
fixtures, and size-independent properties at the bench's full size."""
import os

import numpy as np
import pytest

import oracle_py as orc
import ray_tracing_amd as rt
from helpers import bitwise_equal, default_app, field_app, gpu_render, oracle_scene, oracle_view, parity, soup_app

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _golden(name):
    g = np.load(os.path.join(HERE, "golden", name + ".npz"))
    scene = rt.Scene(g["triangles"], g["meshes"], g["vertices"], g["materials"])
    W, H = (int(x) for x in g["size"])
    return g, scene, W, H


@pytest.mark.parametrize("name", ["default_64x36_shadow", "spheres8_48x32"])
@pytest.mark.parametrize("extra", [0, rt.FLAG_BINARY_TREE, rt.FLAG_WAVE_TRAVERSAL, rt.FLAG_STATIC_GRID, rt.FLAG_STATIC_GRID | rt.FLAG_BINARY_TREE,
                                   rt.FLAG_BRUTE_FORCE])
def test_golden_fixtures(name, extra):
    g, scene, W, H = _golden(name)
    with rt.Context(0) as ctx:
        ctx.upload_scene(scene)
        ctx.set_view(g["world_from_clip"], g["world_position"], W, H)
        flags = (rt.FLAG_SHADOW if int(g["flags"]) & orc.FLAG_SHADOW else 0) | extra
        assert ctx.render(flags=flags) == rt.RT_OK
        assert bitwise_equal(ctx.read_prim_ids(), g["prim"]) and bitwise_equal(ctx.read_hit_t(), g["t"])
        assert bitwise_equal(ctx.read_radiance(), g["rgba"])
        c = ctx.counters()
        assert c.rays_primary + c.rays_shadow == int(g["rays"])
        if "path_accum" in g and extra in (0, rt.FLAG_BINARY_TREE):
            spp = int(g["path_spp"])
            assert ctx.render(flags=rt.FLAG_PATH_TRACE | extra, spp=spp, max_segments=4, seed=int(g["path_seed"])) == rt.RT_OK
            assert bitwise_equal(ctx.read_accumulation(), g["path_accum"])
            c = ctx.counters()
            assert c.rays_primary + c.rays_bounce == int(g["path_segments"])


@pytest.mark.parametrize("segments", [1, 4, 8])
def test_path_tracing_matches_the_oracle(segments):
    app = default_app(96, 54)
    scene = app.prepare_meshinfo()
    osc, view = oracle_scene(scene), oracle_view(app)
    acc_o, segs_o = orc.render_path(osc, view, 3, segments, seed=21, stream=2)
    with rt.Context(0) as ctx:
        ctx.upload_scene(scene)
        m, p = app.camera_view()
        ctx.set_view(m, p, 96, 54)
        assert ctx.render(flags=rt.FLAG_PATH_TRACE | rt.FLAG_COUNTERS, spp=3, max_segments=segments, seed=21, stream_id=2) == rt.RT_OK
        acc_g, rad_g = ctx.read_accumulation(), ctx.read_radiance()
        c = ctx.counters()
        rad_o = orc.resolve(acc_o, 3)
        rmse = float(np.sqrt(np.mean((rad_g.astype(np.float64) - rad_o.astype(np.float64)) ** 2)))
        assert rmse <= 1e-3, rmse                                    # the stated tolerance
        assert bitwise_equal(acc_g, acc_o) and bitwise_equal(rad_g, rad_o)   # and in fact bit-exact (integer RNG, det. log/cos)
        assert c.rays_primary == 96 * 54 * 3 and c.rays_primary + c.rays_bounce == segs_o
        assert c.node_visits > 0


@pytest.mark.parametrize("spp", [4, 5, 21])
def test_path_tracing_sample_batches_match_the_oracle(spp):
    """Samples are traced 16 or 4 at a time on a magnified grid (k_path_start), the rest one by one: 4 = one batch of 4,
    5 = 4 + 1, 21 = 16 + 4 + 1. The running sums must be the sequential ones, bit for bit -- also in a viewport with
    ragged edges, on a tile shard, and across two accumulating frames."""
    vp = (7, 5, 50, 27)
    app = default_app(70, 40, viewport=vp)
    scene = app.prepare_meshinfo()
    osc, view = oracle_scene(scene), oracle_view(app, vp)
    acc_o, segs_o = orc.render_path(osc, view, spp, 4, seed=13, stream=1)
    with rt.Context(0) as ctx:
        ctx.upload_scene(scene)
        m, p = app.camera_view()
        ctx.set_view(m, p, 70, 40, vp)
        assert ctx.render(flags=rt.FLAG_PATH_TRACE, spp=spp, max_segments=4, seed=13, stream_id=1) == rt.RT_OK
        acc_g = ctx.read_accumulation()
        c = ctx.counters()
        assert bitwise_equal(acc_g, acc_o) and bitwise_equal(ctx.read_radiance(), orc.resolve(acc_o, spp))
        assert c.rays_primary == 50 * 27 * spp and c.rays_primary + c.rays_bounce == segs_o
        outside = np.ones((40, 70), bool)
        outside[5:32, 7:57] = False
        assert np.all(acc_g[outside] == 0)                                   # alpha counts samples only where the frame was covered
        # the same samples as two accumulating frames, split off the batch boundaries
        first = max(1, spp // 2 - 1)
        ctx.render(flags=rt.FLAG_PATH_TRACE, spp=first, max_segments=4, seed=13, stream_id=1)
        ctx.render(flags=rt.FLAG_PATH_TRACE | rt.FLAG_ACCUMULATE, spp=spp - first, sample_first=first, max_segments=4, seed=13, stream_id=1)
        assert bitwise_equal(ctx.read_accumulation(), acc_o)
        # a tile shard of it: owned tiles as in the full frame, the others untouched
        from ray_tracing_amd import multi_gpu
        ctx.set_view(m, p, 70, 40)
        ctx.render(flags=rt.FLAG_PATH_TRACE, spp=spp, max_segments=4, seed=13, stream_id=1)
        full = ctx.read_accumulation()
        ctx.render(flags=rt.FLAG_PATH_TRACE, spp=spp, max_segments=4, seed=13, stream_id=1, shard_rank=1, shard_count=3, tile_size=16)
        part = ctx.read_accumulation()
        mine = multi_gpu.tile_owner(70, 40, 16, 3) == 1
        assert bitwise_equal(part[mine], full[mine]) and np.all(part[~mine] == 0)


def test_progressive_accumulation_and_reset_on_camera_move():
    app = default_app(64, 36)
    scene = app.prepare_meshinfo()
    osc, view = oracle_scene(scene), oracle_view(app)
    with rt.Context(0) as ctx:
        ctx.upload_scene(scene)
        m, p = app.camera_view()
        ctx.set_view(m, p, 64, 36)
        # 2 + 3 samples over two frames == 5 samples in one go (running sums are sample-sequential)
        ctx.render(flags=rt.FLAG_PATH_TRACE, spp=2, seed=9)
        ctx.render(flags=rt.FLAG_PATH_TRACE | rt.FLAG_ACCUMULATE, spp=3, sample_first=2, seed=9)
        acc_o, _ = orc.render_path(osc, view, 5, 4, seed=9)
        assert bitwise_equal(ctx.read_accumulation(), acc_o)
        assert bitwise_equal(ctx.read_radiance(), orc.resolve(acc_o, 5))
        # moving the camera restarts the accumulation even with ACCUMULATE set (intent of src/scene.rs:94-96)
        app.fly_cam_step(keys=rt.lib() and 1, dt=0.1)
        m2, p2 = app.camera_view()
        ctx.set_view(m2, p2, 64, 36)
        ctx.render(flags=rt.FLAG_PATH_TRACE | rt.FLAG_ACCUMULATE, spp=2, seed=9)
        acc2, _ = orc.render_path(osc, orc.make_view(m2, p2, 64, 36), 2, 4, seed=9)
        assert bitwise_equal(ctx.read_accumulation(), acc2)


def test_viewport_renders_only_its_rect_with_rebased_uv():
    vp = (16, 8, 64, 36)
    app = default_app(96, 60, viewport=vp)
    scene = app.prepare_meshinfo()
    rgba_o, prim_o, t_o, _ = orc.render_primary(oracle_scene(scene), oracle_view(app, vp), orc.FLAG_SHADOW)
    ctx, rgba_g, prim_g, t_g = gpu_render(app, scene, rt.FLAG_SHADOW, viewport=vp)
    assert bitwise_equal(rgba_g, rgba_o) and bitwise_equal(prim_g, prim_o) and bitwise_equal(t_g, t_o)
    outside = np.ones((60, 96), bool)
    outside[8:44, 16:80] = False
    assert np.all(prim_g[outside] == rt.MISS_PRIM) and np.all(rgba_g[outside] == 0)
    assert ctx.counters().rays_primary == 64 * 36
    ctx.close()


@pytest.mark.parametrize("count,tile", [(2, 64), (3, 16), (8, 32)])
def test_tile_shards_partition_the_frame(count, tile):
    from ray_tracing_amd import multi_gpu
    app = default_app(200, 120)
    scene = app.prepare_meshinfo()
    ctx, full, prim_full, _ = gpu_render(app, scene, rt.FLAG_SHADOW)
    owner = multi_gpu.tile_owner(200, 120, tile, count)
    total = np.zeros_like(full)
    rays = 0
    for rank in range(count):
        _, part, prim_part, _ = gpu_render(app, scene, rt.FLAG_SHADOW, ctx=ctx, shard_rank=rank, shard_count=count, tile_size=tile)
        mine = owner == rank
        assert bitwise_equal(part[mine], full[mine]) and np.all(part[~mine] == 0)
        assert np.all(prim_part[~mine] == rt.MISS_PRIM)
        total += part
        rays += ctx.counters().rays_primary
    assert np.array_equal(total, full) and rays == 200 * 120          # sum of shards == gather of tiles (-0 + 0 = +0: values, not bits)
    ctx.close()


def test_presentation_bgra8_srgb():
    """f2 (src/pipeline.rs:56-60): byte work, so the bytes are the oracle's, all of them -- on a rendered frame and on a
    2^24-value sweep of [0, 1] plus the knee of the OETF, every byte threshold and its predecessor, NaN, < 0 and > 1."""
    app = default_app(64, 36)
    scene = app.prepare_meshinfo()
    ctx, rgba, _, _ = gpu_render(app, scene, rt.FLAG_SHADOW)
    assert bitwise_equal(ctx.read_bgra8_srgb(), orc.encode_bgra8_srgb(rgba))
    n = 1 << 24
    x = np.empty((n + 4096, 4), np.float32)
    x[:n, 0] = np.arange(n, dtype=np.float64) / n                                   # red: every k / 2^24
    x[:n, 1] = np.random.default_rng(1).random(n, dtype=np.float32)                   # green: random
    x[:n, 2] = (np.arange(n, dtype=np.float64) / n) ** 3                              # blue: dense near 0 (the linear piece)
    x[:n, 3] = np.arange(n, dtype=np.float64)[::-1] / n                               # alpha: linear unorm8
    import re
    bits = np.array([int(v, 16) for v in re.findall(r"0x([0-9A-F]{8})u", open(os.path.join(HERE, "..", "oracle", "srgb_table.h")).read())], np.uint32)
    t = bits.view(np.float32)
    special = np.concatenate([t, np.nextafter(t, np.float32(-1)), np.nextafter(t, np.float32(2)),
                              np.array([0.0, -0.0, -1.0, 1.0, 1.5, np.inf, -np.inf, np.nan, 0.0031308, np.nextafter(np.float32(0.0031308), np.float32(1)),
                                        1e-45, 1e-30, 0.5, np.nextafter(np.float32(1), np.float32(0))], np.float32)])
    x[n:] = np.resize(special, (4096, 1))
    assert bitwise_equal(ctx.encode_bgra8_srgb(x), orc.encode_bgra8_srgb(x))
    ctx.close()


def test_presented_frame_to_png_and_ppm(tmp_path):
    """f2: the frame as the reference's Bgra8UnormSrgb target holds it, written where a screenshot of the app can be
    compared with it."""
    import zlib
    app = default_app(96, 54)
    scene = app.prepare_meshinfo()
    ctx, rgba, _, _ = gpu_render(app, scene, rt.FLAG_SHADOW)
    bgra = ctx.read_bgra8_srgb()
    rt.write_image(tmp_path / "frame.png", bgra)
    rt.write_image(tmp_path / "frame.ppm", bgra)
    raw = (tmp_path / "frame.ppm").read_bytes()
    rgb = np.frombuffer(raw[raw.index(b"255\n") + 4:], np.uint8).reshape(54, 96, 3)
    assert np.array_equal(rgb, bgra[..., [2, 1, 0]]) and rgb.max() > 100 and rgb.min() == 0
    png = (tmp_path / "frame.png").read_bytes()
    idat = png[png.index(b"IDAT") + 4:png.index(b"IEND") - 8]
    rows = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(54, 96 * 4 + 1)
    assert np.array_equal(rows[:, 1:].reshape(54, 96, 4)[..., :3], rgb)
    ctx.close()


def test_plugin_lifecycle_and_dirty_upload():
    app = default_app(80, 45)
    assert app.update() == rt.RT_NOT_READY                         # pipeline not created yet: silent skip (src/node.rs:52-56)
    app.finish(0)                                                  # RayTracingPlugin::finish
    assert app.update(flags=rt.FLAG_SHADOW) == rt.RT_OK
    ctx = app.ctx()
    scene = app.info()
    rgba_o, prim_o, _, _ = orc.render_primary(oracle_scene(scene), oracle_view(app), orc.FLAG_SHADOW)
    assert bitwise_equal(ctx.read_radiance(), rgba_o) and bitwise_equal(ctx.read_prim_ids(), prim_o)
    assert (ctx.bvh_info().n_builds, ctx.bvh_info().n_refits) == (1, 0)
    for frame in range(1, 4):
        app.rotate_tagged(0.25)                                    # `rotate` system (src/main.rs:122-126): vertices move
        assert app.update(flags=rt.FLAG_SHADOW) == rt.RT_OK        # re-extracted; records re-baked, BVH refitted (8f item 1)
        scene2 = app.info()
        assert scene2.vertices.tobytes() != scene.vertices.tobytes()
        rgba_o2, prim_o2, _, _ = orc.render_primary(oracle_scene(scene2), oracle_view(app), orc.FLAG_SHADOW)
        assert bitwise_equal(ctx.read_radiance(), rgba_o2) and bitwise_equal(ctx.read_prim_ids(), prim_o2)
        assert (ctx.bvh_info().n_builds, ctx.bvh_info().n_refits) == (1, frame)
    assert app.update_counts() == (3, 0)                           # one entity of six moved: the dirty-mesh path each time
    assert app.update(flags=rt.FLAG_SHADOW) == rt.RT_OK            # nothing moved: neither upload nor refit
    assert (ctx.bvh_info().n_builds, ctx.bvh_info().n_refits) == (1, 3)
    m = app.add_material((0.1, 0.9, 0.1, 1.0))
    app.spawn(app.add_icosphere(0.4, 2), m, translation=(0.0, 1.5, 0.0))   # topology changes: full rebuild
    assert app.update(flags=rt.FLAG_SHADOW) == rt.RT_OK
    assert ctx.bvh_info().n_builds == 2
    scene3 = app.info()
    rgba_o3, prim_o3, _, _ = orc.render_primary(oracle_scene(scene3), oracle_view(app), orc.FLAG_SHADOW)
    assert bitwise_equal(ctx.read_radiance(), rgba_o3) and bitwise_equal(ctx.read_prim_ids(), prim_o3)
    app.camera_set_graph(False)                                    # Tab: camera back on Core3d (src/main.rs:128-147)
    assert app.update() == rt.RT_NOT_READY
    app.camera_set_graph(True)
    # many entities move at once (more than half the scene): the full update; and a material edited in place: a rebuild
    for e in range(4):
        app.set_transform(e, translation=(0.3 * e, 0.2, -0.1 * e))
    assert app.update(flags=rt.FLAG_SHADOW) == rt.RT_OK
    assert app.update_counts() == (3, 1) and ctx.bvh_info().n_builds == 2
    scene4 = app.info()
    rgba_o4, prim_o4, _, _ = orc.render_primary(oracle_scene(scene4), oracle_view(app), orc.FLAG_SHADOW)
    assert bitwise_equal(ctx.read_radiance(), rgba_o4) and bitwise_equal(ctx.read_prim_ids(), prim_o4)
    # finish() again (a device lost and found): a new context, everything uploaded again, the same frame
    app.finish(0)
    assert app.update(flags=rt.FLAG_SHADOW) == rt.RT_OK
    ctx = app.ctx()
    assert ctx.bvh_info().n_builds == 1 and bitwise_equal(ctx.read_radiance(), rgba_o4)
    app.close()


def test_update_vertices_refit_matches_a_rebuild_and_rejects_other_topologies():
    app = soup_app(30, 0x5EED0003, 160, 90, subdivisions=3)
    app.camera_frame_scene(1.0)
    scene = app.prepare_meshinfo()
    with rt.Context(0) as ctx:
        ctx.upload_scene(scene)
        moved = rt.Scene(scene.triangles, scene.meshes.copy(), scene.vertices.copy(), scene.materials)
        rng = np.random.default_rng(3)
        shift = rng.uniform(-3, 3, size=(len(moved.meshes), 3)).astype(np.float32)      # every sphere drifts
        for k, mi in enumerate(moved.meshes):
            tri = moved.triangles["indices"][mi["first_tri"]:mi["first_tri"] + mi["tri_count"]]
            vs = np.unique(tri)
            moved.vertices["pos"][vs] += shift[k]
            moved.meshes["aabb_min"][k] += shift[k]
            moved.meshes["aabb_max"][k] += shift[k]
        ctx.update_vertices(moved)
        m, p = app.camera_view()
        ctx.set_view(m, p, 160, 90)
        assert ctx.render(flags=rt.FLAG_SHADOW) == rt.RT_OK
        rgba_o, prim_o, t_o, _ = orc.render_primary(oracle_scene(moved), oracle_view(app), orc.FLAG_SHADOW)
        assert bitwise_equal(ctx.read_prim_ids(), prim_o) and bitwise_equal(ctx.read_hit_t(), t_o)
        assert bitwise_equal(ctx.read_radiance(), rgba_o)
        assert ctx.bvh_info().n_refits == 1
        # the GPU refit (one launch per tree level) gives the boxes and records of the host refit, bit for bit
        _, nodes_h, recs_h = rt.host_bvh_build(scene, threads=2, refit_to=moved)
        nodes_g, recs_g = ctx.read_bvh()
        assert bitwise_equal(nodes_g, nodes_h) and bitwise_equal(recs_g, recs_h)
        # ... and the 4-wide tree re-quantised on the device equals the one re-quantised on the host (same topology,
        # the boxes of the refitted binary tree): a second context takes the host refit path
        wide_g = ctx.read_bvh_wide()
        assert len(wide_g) == ctx.bvh_info().n_wide_nodes > 0
        os.environ["RT_TUNE_HOST_REFIT"] = "1"
        try:
            with rt.Context(0) as ctx2:
                ctx2.upload_scene(scene)
                wide_0 = ctx2.read_bvh_wide()
                ctx2.update_vertices(moved)
                wide_h = ctx2.read_bvh_wide()
        finally:
            del os.environ["RT_TUNE_HOST_REFIT"]
        assert bitwise_equal(wide_g, wide_h) and not bitwise_equal(wide_h, wide_0)
        assert np.array_equal(wide_h["child"], wide_0["child"])                   # topology kept
        assert ctx.render(flags=rt.FLAG_SHADOW | rt.FLAG_BINARY_TREE) == rt.RT_OK # both trees give the oracle's frame
        assert bitwise_equal(ctx.read_prim_ids(), prim_o) and bitwise_equal(ctx.read_radiance(), rgba_o)
        # vertices that leave the wide step's coordinate range (RT_WIDE_COORD_MAX): the same frame from the binary tree
        far = rt.Scene(scene.triangles, moved.meshes.copy(), moved.vertices.copy(), scene.materials)
        far.vertices["pos"][:, 0] += np.float32(1e8)
        far.meshes["aabb_min"][:, 0] += np.float32(1e8)
        far.meshes["aabb_max"][:, 0] += np.float32(1e8)
        ctx.update_vertices(far)
        assert ctx.render(flags=rt.FLAG_SHADOW | rt.FLAG_COUNTERS) == rt.RT_OK
        v_far = ctx.counters().node_visits
        prim_far = ctx.read_prim_ids()
        assert ctx.render(flags=rt.FLAG_SHADOW | rt.FLAG_COUNTERS | rt.FLAG_BINARY_TREE) == rt.RT_OK
        assert ctx.counters().node_visits == v_far and bitwise_equal(ctx.read_prim_ids(), prim_far)
        ctx.update_vertices(moved)                                                 # ... and back in range: wide again
        assert ctx.render(flags=rt.FLAG_SHADOW | rt.FLAG_COUNTERS) == rt.RT_OK
        assert ctx.counters().node_visits < 0.75 * v_far * 10 and bitwise_equal(ctx.read_prim_ids(), prim_o)
        assert bitwise_equal(ctx.read_bvh()[0], nodes_h) and ctx.bvh_info().n_refits == 3
        bad = rt.Scene(scene.triangles, moved.meshes, moved.vertices.copy(), scene.materials)
        bad.vertices["pos"][int(moved.triangles["indices"][0][0])] = np.nan       # a non-finite vertex needs a rebuild
        with pytest.raises(rt.RtError) as e:
            ctx.update_vertices(bad)
        assert e.value.status == rt.RT_ERR_SCENE
        assert bitwise_equal(ctx.read_bvh()[0], nodes_h)                          # ... and left the tree untouched
        fewer = rt.Scene(scene.triangles, scene.meshes[:-1], scene.vertices, scene.materials)
        with pytest.raises(rt.RtError) as e:
            ctx.update_vertices(fewer)
        assert e.value.status == rt.RT_ERR_SCENE


@pytest.mark.parametrize("kind", ["default", "soup", "analytic"])
def test_wide_and_binary_trees_give_identical_results(kind):
    """The 4-wide quantised tree (default) and the binary tree it was collapsed from (RT_FLAG_BINARY_TREE): the same
    primitive, distance and radiance in every pixel, for every kernel that walks a tree, with about half the node visits."""
    if kind == "default":
        app = default_app(320, 180)
        scene = app.prepare_meshinfo()
    elif kind == "soup":
        app = soup_app(60, 0x5EED0003, 320, 180, subdivisions=3)
        app.camera_frame_scene(1.0)
        scene = app.prepare_meshinfo()
    else:
        scene = rt.analytic_sphere_field(500, 0x5EED0002)
        app = default_app(320, 180)
        app.camera_look_at((60.0, 30.0, 60.0), (0, 5, 0), (0, 1, 0))
    m, p = app.camera_view()
    with rt.Context(0) as ctx:
        ctx.upload_scene(scene)
        ctx.set_view(m, p, 320, 180)
        assert ctx.bvh_info().n_wide_nodes > 0
        for flags in (rt.FLAG_SHADOW, rt.FLAG_SHADOW | rt.FLAG_STATIC_GRID, 0):
            out = []
            for tree in (0, rt.FLAG_BINARY_TREE):
                assert ctx.render(flags=flags | tree | rt.FLAG_COUNTERS) == rt.RT_OK
                out.append((ctx.read_prim_ids(), ctx.read_hit_t(), ctx.read_radiance(), ctx.counters()))
            assert all(bitwise_equal(out[0][i], out[1][i]) for i in range(3))
            assert out[0][3].rays_primary == out[1][3].rays_primary and out[0][3].rays_shadow == out[1][3].rays_shadow
            assert out[0][3].node_visits < 0.75 * out[1][3].node_visits
        acc = []
        for tree in (0, rt.FLAG_BINARY_TREE):
            assert ctx.render(flags=rt.FLAG_PATH_TRACE | tree, spp=2, max_segments=4, seed=5) == rt.RT_OK
            acc.append((ctx.read_accumulation(), ctx.counters()))
        assert bitwise_equal(acc[0][0], acc[1][0]) and acc[0][1].rays_bounce == acc[1][1].rays_bounce


@pytest.mark.parametrize("case", ["far_camera", "tiny_scene_far_from_origin", "large_scene", "huge_coordinates", "huge_camera"])
def test_wide_tree_conservative_culling_at_the_extremes(case):
    """Where the wide step's byte*A - B form is least comfortable: an origin thousands of scene sizes away (its rounding
    error scales with the distance), a small scene far from the world origin (the grid expansion scales with the
    coordinates), a scene of millions of units, and -- beyond RT_WIDE_COORD_MAX, where the wide step's terms could overflow --
    coordinates near FLT_MAX and a camera 1e9 away, which the library walks on the binary tree. Oracle == wide == binary."""
    if case == "far_camera":
        app = default_app(48, 48)
        app.camera_set_projection(0.0008, 0.1)
        app.camera_look_at((9000.0, 6000.0, 12000.0), (0, 0.5, 0), (0, 1, 0))
        scene = app.prepare_meshinfo()
    elif case == "tiny_scene_far_from_origin":
        app = default_app(48, 48)
        base = app.prepare_meshinfo()
        v = base.vertices.copy()
        v["pos"] = v["pos"] * np.float32(1e-3) + np.float32([4096.0, -2048.0, 1024.0])
        m = base.meshes.copy()
        m["aabb_min"] = m["aabb_min"] * np.float32(1e-3) + np.float32([4096.0, -2048.0, 1024.0])
        m["aabb_max"] = m["aabb_max"] * np.float32(1e-3) + np.float32([4096.0, -2048.0, 1024.0])
        scene = rt.Scene(base.triangles, m, v, base.materials)
        app.camera_look_at((4096.0, -2047.997, 1024.005), (4096.0, -2048.0, 1024.0), (0, 1, 0))
    elif case in ("large_scene", "huge_camera"):
        app = default_app(48, 48)
        base = app.prepare_meshinfo()
        v = base.vertices.copy()
        v["pos"] = v["pos"] * np.float32(1e6)
        m = base.meshes.copy()
        m["aabb_min"] = m["aabb_min"] * np.float32(1e6)
        m["aabb_max"] = m["aabb_max"] * np.float32(1e6)
        scene = rt.Scene(base.triangles, m, v, base.materials)
        if case == "large_scene":
            app.camera_look_at((0.0, 3e6, 5e6), (0, 0, 0), (0, 1, 0))
        else:
            app.camera_set_projection(0.06, 0.1)
            app.camera_look_at((1.2e8, 0.8e8, 1.6e8), (0, 5e5, 0), (0, 1, 0))
    else:
        app = default_app(48, 48)
        base = app.prepare_meshinfo()
        nv = len(base.vertices)
        v = np.concatenate([base.vertices, np.zeros(6, rt.VERTEX)])
        v["pos"][nv:] = [(-3e38, 0, 0), (-3e38, 1, 0), (-3e38, 0, 1), (3e38, 0, 0), (3e38, 1, 0), (3e38, 0, 1)]
        v["norm"][nv:] = (0, 1, 0)
        t = np.concatenate([base.triangles, np.zeros(2, rt.TRIANGLE)])
        t["indices"][-2:] = [(nv, nv + 1, nv + 2), (nv + 3, nv + 4, nv + 5)]
        m = np.concatenate([base.meshes, np.zeros(1, rt.MESH_INFO)])
        m["first_tri"][-1] = len(base.triangles)
        m["tri_count"][-1] = 2
        m["aabb_min"][-1] = (-3e38, 0, 0)
        m["aabb_max"][-1] = (3e38, 1, 1)
        scene = rt.Scene(t, m, v, base.materials)
    rgba_o, prim_o, t_o, _ = orc.render_primary(oracle_scene(scene), oracle_view(app), orc.FLAG_SHADOW)
    assert (prim_o != rt.MISS_PRIM).mean() > 0.2                     # the view does look at the scene
    for tree in (0, rt.FLAG_BINARY_TREE):
        ctx, rgba_g, prim_g, t_g = gpu_render(app, scene, rt.FLAG_SHADOW | tree | rt.FLAG_COUNTERS)
        assert (ctx.bvh_info().n_wide_nodes > 0) == (case != "huge_coordinates")
        if tree == 0:
            visits_wide = ctx.counters().node_visits
        elif case in ("huge_coordinates", "huge_camera"):
            assert ctx.counters().node_visits == visits_wide            # both renders walked the binary tree
        else:
            assert ctx.counters().node_visits > visits_wide
        assert bitwise_equal(prim_g, prim_o) and bitwise_equal(t_g, t_o) and bitwise_equal(rgba_g, rgba_o)
        ctx.close()


@pytest.mark.parametrize("list_max,levels_max", [("16384", "262144"), ("0", "0"), ("700", "262144"), ("0", "262144"), ("700", "0")])
def test_update_vertex_range_refits_only_the_moved_meshes(list_max, levels_max, monkeypatch):
    """f1, dirty-mesh form: four of thirty spheres move; only their vertices are sent, only their leaves re-baked and only
    their ancestors refitted -- and the tree (binary and wide) is the one a full rt_update_vertices leaves, bit for bit.
    All three forms of the device side: one launch of one workgroup over a host-made list of the nodes above the moved
    meshes (default for short lists), one launch per tree level over that list's nodes of the level (lists beyond
    RT_TUNE_REFIT_LIST_MAX, up to RT_TUNE_REFIT_LEVELS_MAX), and the per-level launches over the whole tree with a dirty
    byte per node (beyond both limits)."""
    monkeypatch.setenv("RT_TUNE_REFIT_LIST_MAX", list_max)
    monkeypatch.setenv("RT_TUNE_REFIT_LEVELS_MAX", levels_max)
    app = soup_app(30, 0x5EED0003, 160, 90, subdivisions=3)
    app.camera_frame_scene(1.0)
    scene = app.prepare_meshinfo()
    m0, m1 = 5, 9
    moved = rt.Scene(scene.triangles, scene.meshes.copy(), scene.vertices.copy(), scene.materials)
    rng = np.random.default_rng(11)
    v_lo, v_hi = 2**32, 0
    for k in range(m0, m1):
        mi = moved.meshes[k]
        vs = np.unique(moved.triangles["indices"][mi["first_tri"]:mi["first_tri"] + mi["tri_count"]])
        shift = rng.uniform(-4, 4, 3).astype(np.float32)
        moved.vertices["pos"][vs] += shift
        moved.meshes["aabb_min"][k] += shift
        moved.meshes["aabb_max"][k] += shift
        v_lo, v_hi = min(v_lo, int(vs.min())), max(v_hi, int(vs.max()))
    m, p = app.camera_view()
    rgba_o, prim_o, t_o, _ = orc.render_primary(oracle_scene(moved), oracle_view(app), orc.FLAG_SHADOW)
    with rt.Context(0) as full, rt.Context(0) as part:
        for ctx in (full, part):
            ctx.upload_scene(scene)
            ctx.set_view(m, p, 160, 90)
        full.update_vertices(moved)
        part.update_vertex_range(moved, v_lo, v_hi - v_lo + 1, m0, m1 - m0)
        assert part.bvh_info().n_refits == 1
        for tree in (0, rt.FLAG_BINARY_TREE):
            assert part.render(flags=rt.FLAG_SHADOW | tree) == rt.RT_OK
            assert bitwise_equal(part.read_prim_ids(), prim_o) and bitwise_equal(part.read_hit_t(), t_o)
            assert bitwise_equal(part.read_radiance(), rgba_o)
        nodes_f, recs_f = full.read_bvh()
        nodes_p, recs_p = part.read_bvh()
        assert bitwise_equal(nodes_p, nodes_f) and bitwise_equal(recs_p, recs_f)
        assert bitwise_equal(part.read_bvh_wide(), full.read_bvh_wide())
        # a second, different move of the same meshes, then back: still the full update's tree.  The same range again is
        # what an animation does: a long list is made on the first repeat (until then the whole-tree launches serve) and
        # every list is kept on the device for the calls that follow
        part.update_vertex_range(scene, v_lo, v_hi - v_lo + 1, m0, m1 - m0)
        full.update_vertices(scene)
        assert bitwise_equal(part.read_bvh()[0], full.read_bvh()[0]) and bitwise_equal(part.read_bvh_wide(), full.read_bvh_wide())
        for pose in (moved, scene, moved):
            part.update_vertex_range(pose, v_lo, v_hi - v_lo + 1, m0, m1 - m0)
            full.update_vertices(pose)
            assert bitwise_equal(part.read_bvh()[0], full.read_bvh()[0]) and bitwise_equal(part.read_bvh()[1], full.read_bvh()[1])
            assert bitwise_equal(part.read_bvh_wide(), full.read_bvh_wide())
        # another range in between replaces the kept list; back on the first range a new one is made
        other = rt.Scene(scene.triangles, moved.meshes.copy(), moved.vertices.copy(), scene.materials)
        mo = other.meshes[20]
        vo = np.unique(other.triangles["indices"][mo["first_tri"]:mo["first_tri"] + mo["tri_count"]])
        other.vertices["pos"][vo] += np.float32(0.5); other.meshes["aabb_min"][20] += np.float32(0.5); other.meshes["aabb_max"][20] += np.float32(0.5)
        for _ in range(2):
            part.update_vertex_range(other, int(vo.min()), int(vo.max() - vo.min() + 1), 20, 1)
        part.update_vertex_range(other, v_lo, v_hi - v_lo + 1, m0, m1 - m0)
        full.update_vertices(other)
        assert bitwise_equal(part.read_bvh()[0], full.read_bvh()[0]) and bitwise_equal(part.read_bvh_wide(), full.read_bvh_wide())
        part.update_vertex_range(moved, int(vo.min()), int(vo.max() - vo.min() + 1), 20, 1)
        full.update_vertices(moved)
        assert bitwise_equal(part.read_bvh()[0], full.read_bvh()[0]) and bitwise_equal(part.read_bvh_wide(), full.read_bvh_wide())
        # the range must hold exactly the moved meshes' vertices
        with pytest.raises(rt.RtError) as e:
            part.update_vertex_range(moved, v_lo + 1, v_hi - v_lo, m0, m1 - m0)        # misses a vertex of mesh m0
        assert e.value.status == rt.RT_ERR_SCENE
        with pytest.raises(rt.RtError) as e:
            part.update_vertex_range(moved, v_lo, v_hi - v_lo + 2, m0, m1 - m0)        # reaches into mesh m1's vertices
        assert e.value.status == rt.RT_ERR_SCENE
        with pytest.raises(rt.RtError) as e:
            part.update_vertex_range(moved, v_lo, v_hi - v_lo + 1, m0, len(scene.meshes))   # mesh range out of bounds
        assert e.value.status == rt.RT_ERR_SCENE
        bad = rt.Scene(scene.triangles, moved.meshes, moved.vertices.copy(), scene.materials)
        bad.vertices["pos"][v_lo + 3] = np.inf
        with pytest.raises(rt.RtError) as e:
            part.update_vertex_range(bad, v_lo, v_hi - v_lo + 1, m0, m1 - m0)
        assert e.value.status == rt.RT_ERR_SCENE
        assert bitwise_equal(part.read_bvh()[0], full.read_bvh()[0])                    # refused calls changed nothing


@pytest.mark.parametrize("seed", range(8))
def test_fuzz_wide_tree_binary_tree_and_reference_loop_agree(seed):
    """Random triangle soups over many scales (slivers, coplanar sheets, coincident and degenerate triangles, a few huge
    ones around thousands of tiny ones), random and axis-parallel cameras: the wide tree, the binary tree and the
    reference-shaped two-level loop (RT_FLAG_BRUTE_FORCE, itself pinned to the oracle elsewhere) must give the same
    primitive, distance and radiance in every pixel, shadow rays included."""
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(200, 3000))
    scale = float(10.0 ** rng.uniform(-3, 4))
    centre = rng.uniform(-1, 1, 3) * scale * float(rng.choice([0.0, 1.0, 100.0]))
    c = rng.uniform(-1, 1, (n, 1, 3)) * scale
    size = (10.0 ** rng.uniform(-3, 0, (n, 1, 1))) * scale
    tri = c + rng.normal(size=(n, 3, 3)) * size
    kind = rng.integers(0, 10, n)
    tri[kind == 0, :, 1] = 0.0                                   # a coplanar sheet at y = 0
    tri[kind == 1, 2] = tri[kind == 1, 1]                        # degenerate: two equal vertices
    tri[kind == 2] = tri[np.flatnonzero(kind == 2)[:1]] if (kind == 2).any() else tri[kind == 2]   # coincident copies
    tri[kind == 3, :, 0] = np.round(tri[kind == 3, :, 0] / scale) * scale    # axis-aligned slivers on grid planes
    tri += centre
    verts = np.zeros(3 * n, rt.VERTEX)
    verts["pos"] = tri.reshape(-1, 3).astype(np.float32)
    nrm = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
    nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-30)
    verts["norm"] = np.repeat(nrm, 3, axis=0).astype(np.float32)
    tris = np.zeros(n, rt.TRIANGLE)
    tris["indices"] = np.arange(3 * n, dtype=np.uint32).reshape(n, 3)
    n_mesh = int(rng.integers(1, 6))
    cuts = np.sort(rng.integers(0, n + 1, n_mesh - 1))
    starts = np.concatenate([[0], cuts]).astype(np.uint32)
    ends = np.concatenate([cuts, [n]]).astype(np.uint32)
    meshes = np.zeros(n_mesh, rt.MESH_INFO)
    meshes["first_tri"], meshes["tri_count"] = starts, ends - starts
    meshes["material"] = np.arange(n_mesh) % 3
    for k in range(n_mesh):
        pts = verts["pos"][3 * starts[k]:3 * ends[k]]
        if len(pts):
            meshes["aabb_min"][k], meshes["aabb_max"][k] = pts.min(0), pts.max(0)
    if n_mesh > 1 and seed % 2:                                   # the reference's quirk: a box that does not bound its mesh
        meshes["aabb_max"][0] = meshes["aabb_min"][0] + (meshes["aabb_max"][0] - meshes["aabb_min"][0]) * np.float32(0.5)
    mats = np.zeros(3, rt.MATERIAL)
    mats["color"] = [(0.9, 0.2, 0.2, 1), (0.2, 0.9, 0.2, 1), (0.2, 0.2, 0.9, 1)]
    scene = rt.Scene(tris, meshes, verts, mats)
    app = default_app(64, 64)
    eye = centre + rng.normal(size=3) * scale * float(rng.choice([0.5, 2.0, 30.0]))
    if seed % 3 == 0:
        eye = centre + np.array([0.0, 3.0 * scale, 1e-6 * scale])          # straight down: axis-parallel primary rays
    app.camera_look_at(tuple(eye), tuple(centre), (0, 0, -1) if seed % 3 == 0 else (0, 1, 0))
    m, p = app.camera_view()
    with rt.Context(0) as ctx:
        ctx.upload_scene(scene)
        ctx.set_view(m, p, 64, 64)
        out = []
        for flags in (rt.FLAG_SHADOW, rt.FLAG_SHADOW | rt.FLAG_BINARY_TREE, rt.FLAG_SHADOW | rt.FLAG_BRUTE_FORCE,
                      rt.FLAG_SHADOW | rt.FLAG_STATIC_GRID):
            assert ctx.render(flags=flags) == rt.RT_OK
            out.append((ctx.read_prim_ids(), ctx.read_hit_t(), ctx.read_radiance()))
        assert (out[0][0] != rt.MISS_PRIM).mean() > 0.01           # the camera does see the soup
        for o in out[1:]:
            assert bitwise_equal(o[0], out[0][0]) and bitwise_equal(o[1], out[0][1])
            assert np.array_equal(np.isnan(o[2]), np.isnan(out[0][2])) and np.array_equal(np.nan_to_num(o[2]), np.nan_to_num(out[0][2]))
        acc = []
        for tree in (0, rt.FLAG_BINARY_TREE):
            assert ctx.render(flags=rt.FLAG_PATH_TRACE | tree, spp=5, max_segments=3, seed=seed) == rt.RT_OK
            acc.append(ctx.read_accumulation())
        assert np.array_equal(np.isnan(acc[0]), np.isnan(acc[1])) and np.array_equal(np.nan_to_num(acc[0]), np.nan_to_num(acc[1]))
        # the oracle's scalar walk of the same trees: the same hits, and the kernel's counted visits to the unit
        osc, view = oracle_scene(scene), orc.make_view(m, p, 64, 64)
        nodes, recs = ctx.read_bvh()
        wide = ctx.read_bvh_wide() if ctx.bvh_info().n_wide_nodes else None
        for tree_flag, tree in ((0, orc.Tree(nodes, recs, wide)), (rt.FLAG_BINARY_TREE, orc.Tree(nodes, recs))):
            rgba_w, prim_w, t_w, rays_w, cnt_w = orc.render_primary_tree(osc, tree, view, orc.FLAG_SHADOW)
            assert bitwise_equal(prim_w, out[0][0]) and bitwise_equal(t_w, out[0][1])
            assert ctx.render(flags=rt.FLAG_SHADOW | rt.FLAG_COUNTERS | tree_flag) == rt.RT_OK
            c = ctx.counters()
            assert (c.node_visits, c.tri_tests, c.mesh_aabb_tests) == cnt_w and c.rays_primary + c.rays_shadow == rays_w


def test_edge_cases_empty_scene_not_ready_and_bad_arguments():
    with rt.Context(0) as ctx:
        assert ctx.render() == rt.RT_NOT_READY                     # no scene, no view
        empty = rt.Scene(np.zeros(0, rt.TRIANGLE), np.zeros(0, rt.MESH_INFO), np.zeros(0, rt.VERTEX), np.zeros(0, rt.MATERIAL))
        ctx.upload_scene(empty)
        assert ctx.render() == rt.RT_NOT_READY                     # still no view
        ctx.set_view(np.eye(4, dtype=np.float32).ravel(), (0, 0, 0), 33, 17)     # ragged size (not a tile multiple)
        for f in (0, rt.FLAG_SHADOW, rt.FLAG_BRUTE_FORCE, rt.FLAG_WAVE_TRAVERSAL, rt.FLAG_STATIC_GRID, rt.FLAG_PATH_TRACE):
            assert ctx.render(flags=f, spp=1) == rt.RT_OK
            assert np.all(ctx.read_prim_ids() == rt.MISS_PRIM)
        assert np.allclose(ctx.read_radiance()[..., :3], [0.5, 0.71, 0.86])       # path tracing: every path sees the sky
        with pytest.raises(rt.RtError):
            ctx.set_view(np.eye(4, dtype=np.float32).ravel(), (0, 0, 0), 0, 10)
        with pytest.raises(rt.RtError):
            ctx.set_view(np.eye(4, dtype=np.float32).ravel(), (0, 0, 0), 10, 10, (5, 5, 10, 10))
        with pytest.raises(rt.RtError):
            ctx.render(shard_rank=3, shard_count=2)
        bad = rt.Scene(np.zeros(1, rt.TRIANGLE), np.zeros(0, rt.MESH_INFO), np.zeros(0, rt.VERTEX), np.zeros(0, rt.MATERIAL))
        with pytest.raises(rt.RtError) as e:
            ctx.upload_scene(bad)                                   # index 0 with no vertices
        assert e.value.status == rt.RT_ERR_SCENE


def test_degenerate_geometry_nan_vertices_and_zero_normals():
    app = default_app(64, 36)
    scene = app.prepare_meshinfo()
    v = scene.vertices.copy()
    v["pos"][30] = (np.nan, 0, 0)            # a plane-adjacent vertex of the second mesh? any vertex: its triangles vanish
    v["norm"][100:110] = 0                   # zero normals -> NaN shading, same NaN on both sides
    v["pos"][200] = v["pos"][201]            # zero-area triangle(s)
    s2 = rt.Scene(scene.triangles, scene.meshes, v, scene.materials)
    rgba_o, prim_o, t_o, _ = orc.render_primary(oracle_scene(s2), oracle_view(app), orc.FLAG_SHADOW)
    for extra in (0, rt.FLAG_BRUTE_FORCE, rt.FLAG_WAVE_TRAVERSAL):
        ctx, rgba_g, prim_g, t_g = gpu_render(app, s2, rt.FLAG_SHADOW | extra)
        assert bitwise_equal(prim_g, prim_o) and bitwise_equal(t_g, t_o)
        assert np.array_equal(np.isnan(rgba_g), np.isnan(rgba_o))
        assert np.array_equal(np.nan_to_num(rgba_g), np.nan_to_num(rgba_o))
        ctx.close()


def test_axis_parallel_rays_and_rays_starting_on_geometry():
    """Shadow rays are exactly (0,1,0) (two zero components, infinite reciprocals) and start 1e-3 off a
    surface; path segments start ON a triangle. An orthographic-like top-down camera adds primary rays
    with zero components. All must agree with the oracle bit for bit."""
    app = default_app(48, 48)
    app.camera_look_at((0.0, 8.0, 1e-7), (0, 0, 0), (0, 0, -1))
    scene = app.prepare_meshinfo()
    rgba_o, prim_o, t_o, _ = orc.render_primary(oracle_scene(scene), oracle_view(app), orc.FLAG_SHADOW)
    ctx, rgba_g, prim_g, t_g = gpu_render(app, scene, rt.FLAG_SHADOW)
    assert bitwise_equal(prim_g, prim_o) and bitwise_equal(t_g, t_o) and bitwise_equal(rgba_g, rgba_o)
    ctx.close()


def test_full_size_properties_on_the_bench_workload():
    """At the bench's full size the oracle is too slow, so size-independent properties are checked:
    (1) BVH path == reference-shaped brute-force loop on a random pixel sample (oracle on the sample),
    (2) per-lane, wave-coherent and persistent kernels agree on the whole 1080p frame, bit for bit,
    (3) the four tile shards of the frame add up to the frame, (4) counters are consistent; and, since the oracle can
    walk the tree too (orc_hit_tree), (5) the whole frame and the counted visits against that walk."""
    app = field_app(10000, 0x5EED0002, 1920, 1080)
    app.camera_frame_scene(1.0)
    scene = app.prepare_meshinfo()
    ctx, rgba, prim, t = gpu_render(app, scene, rt.FLAG_SHADOW | rt.FLAG_COUNTERS)
    c = ctx.counters()
    hit = prim != rt.MISS_PRIM
    assert c.rays_primary == 1920 * 1080 and c.rays_shadow == int(hit.sum())
    assert 0.5 < hit.mean() < 0.8
    assert np.all(rgba[~hit] == 0) and np.all(t[~hit] == 0) and np.all(t[hit] > 0)
    # every reported primitive belongs to the scene and its mesh's material shades the pixel's colour family
    assert prim[hit].max() < len(scene.triangles)
    rng = np.random.default_rng(5)
    pix = rng.choice(1920 * 1080, 1500, replace=False).astype(np.uint32)
    rgba_o, prim_o, t_o, _ = orc.render_primary(oracle_scene(scene), oracle_view(app), orc.FLAG_SHADOW, threads=16, pixel_list=pix)
    agree, rmse = parity(rgba.reshape(-1, 4)[pix], prim.ravel()[pix], rgba_o, prim_o)
    assert agree >= 0.999 and rmse <= 1e-3, (agree, rmse)
    assert bitwise_equal(prim.ravel()[pix], prim_o) and bitwise_equal(t.ravel()[pix], t_o)
    for extra in (rt.FLAG_WAVE_TRAVERSAL, rt.FLAG_STATIC_GRID):
        _, rgba2, prim2, t2 = gpu_render(app, scene, rt.FLAG_SHADOW | extra, ctx=ctx)
        assert bitwise_equal(prim2, prim) and bitwise_equal(t2, t) and bitwise_equal(rgba2, rgba)
    # (5) the whole frame and the counted visits against the oracle's walk of the device's own tree
    tree = orc.Tree(*ctx.read_bvh(), ctx.read_bvh_wide())
    rgba_w, prim_w, t_w, rays_w, cnt_w = orc.render_primary_tree(oracle_scene(scene), tree, oracle_view(app), orc.FLAG_SHADOW, threads=16)
    assert bitwise_equal(prim, prim_w) and bitwise_equal(t, t_w) and bitwise_equal(rgba, rgba_w)
    assert (c.node_visits, c.tri_tests, c.mesh_aabb_tests) == cnt_w and c.rays_primary + c.rays_shadow == rays_w
    total = np.zeros_like(rgba)
    for rank in range(4):
        _, part, _, _ = gpu_render(app, scene, rt.FLAG_SHADOW, ctx=ctx, shard_rank=rank, shard_count=4)
        total += part
    assert np.array_equal(total, rgba)
    ctx.close()


def test_soup_1m_triangles_sampled_parity():
    app = soup_app(1389, 0x5EED0003, 1920, 1080)
    app.camera_frame_scene(1.0)
    scene = app.prepare_meshinfo()
    assert len(scene.triangles) == 1000080 and len(scene.vertices) == 502818
    ctx, rgba, prim, t = gpu_render(app, scene, rt.FLAG_SHADOW)
    rng = np.random.default_rng(6)
    pix = rng.choice(1920 * 1080, 4000, replace=False).astype(np.uint32)
    rgba_o, prim_o, t_o, _ = orc.render_primary(oracle_scene(scene), oracle_view(app), orc.FLAG_SHADOW, threads=16, pixel_list=pix)
    agree, rmse = parity(rgba.reshape(-1, 4)[pix], prim.ravel()[pix], rgba_o, prim_o)
    assert agree >= 0.999 and rmse <= 1e-3, (agree, rmse)
    # the WHOLE frame against the oracle's walk of the device's own tree (oracle/rt_oracle.c orc_hit_tree; the
    # reference-shaped loop above pins that walk on the sample): results and counted visits
    osc, view = oracle_scene(scene), oracle_view(app)
    tree = orc.Tree(*ctx.read_bvh(), ctx.read_bvh_wide())
    rgba_w, prim_w, t_w, rays_w, cnt_w = orc.render_primary_tree(osc, tree, view, orc.FLAG_SHADOW, threads=16)
    assert bitwise_equal(prim, prim_w) and bitwise_equal(t, t_w) and bitwise_equal(rgba, rgba_w)
    gpu_render(app, scene, rt.FLAG_SHADOW | rt.FLAG_COUNTERS, ctx=ctx)
    c = ctx.counters()
    assert (c.node_visits, c.tri_tests, c.mesh_aabb_tests) == cnt_w and c.rays_primary + c.rays_shadow == rays_w
    # path tracing on the big scene (configs[3]'s scene and loop shape): 1 spp, 4 segments, whole frame, bit for bit
    assert ctx.render(flags=rt.FLAG_PATH_TRACE | rt.FLAG_COUNTERS, spp=1, max_segments=4, seed=1) == rt.RT_OK
    acc = ctx.read_accumulation()
    assert np.all(acc[..., 3] == 1.0) and np.isfinite(acc).all()
    acc_w, segs_w, cnt_p = orc.render_path_tree(osc, tree, view, spp=1, max_segments=4, seed=1, threads=16)
    c = ctx.counters()
    assert bitwise_equal(acc, acc_w)
    assert c.rays_primary + c.rays_bounce == segs_w and (c.node_visits, c.tri_tests, c.mesh_aabb_tests) == cnt_p
    # configs[2]: 4 jittered primary rays per pixel (one batch of four on the magnified grid)
    assert ctx.render(flags=rt.FLAG_PATH_TRACE, spp=4, max_segments=1, seed=3) == rt.RT_OK
    acc4_w, _, _ = orc.render_path_tree(osc, tree, view, spp=4, max_segments=1, seed=3, threads=16)
    assert bitwise_equal(ctx.read_accumulation(), acc4_w)
    # configs[3] at its resolution: 3840x2160, <= 4 segments, 4 of its 64 samples per pixel (one batch of four)
    app.camera_set_target(3840, 2160)
    m, p = app.camera_view()
    ctx.set_view(m, p, 3840, 2160)
    assert ctx.render(flags=rt.FLAG_PATH_TRACE, spp=4, max_segments=4, seed=9) == rt.RT_OK
    acc4k_w, segs4k, _ = orc.render_path_tree(osc, tree, oracle_view(app), spp=4, max_segments=4, seed=9, threads=16)
    c = ctx.counters()
    assert bitwise_equal(ctx.read_accumulation(), acc4k_w) and c.rays_primary + c.rays_bounce == segs4k
    ctx.close()


def test_soup_10m_triangles_path_tracing_whole_frame():
    """configs[4]'s scene (13,889 icospheres = 10,000,080 triangles) and loop shape (<= 8 segments) at 1080p: the whole
    frame, bit for bit, against the oracle's walk of the device's own tree, and the counted visits to the unit."""
    app = soup_app(13889, 0x5EED0005, 1920, 1080)
    app.camera_frame_scene(1.0)
    scene = app.prepare_meshinfo()
    assert len(scene.triangles) == 10000080
    osc, view = oracle_scene(scene), oracle_view(app)
    with rt.Context(0) as ctx:
        ctx.upload_scene(scene)
        m, p = app.camera_view()
        ctx.set_view(m, p, 1920, 1080)
        assert ctx.bvh_info().n_wide_nodes > 0
        tree = orc.Tree(*ctx.read_bvh(), ctx.read_bvh_wide())
        assert ctx.render(flags=rt.FLAG_PATH_TRACE | rt.FLAG_COUNTERS, spp=1, max_segments=8, seed=1) == rt.RT_OK
        acc, c = ctx.read_accumulation(), ctx.counters()
        acc_w, segs_w, cnt_w = orc.render_path_tree(osc, tree, view, spp=1, max_segments=8, seed=1, threads=16)
        assert bitwise_equal(acc, acc_w)
        assert c.rays_primary + c.rays_bounce == segs_w and (c.node_visits, c.tri_tests, c.mesh_aabb_tests) == cnt_w
        # primary + shadow on the same scene
        assert ctx.render(flags=rt.FLAG_SHADOW) == rt.RT_OK
        rgba_w, prim_w, t_w, _, _ = orc.render_primary_tree(osc, tree, view, orc.FLAG_SHADOW, threads=16)
        assert bitwise_equal(ctx.read_prim_ids(), prim_w) and bitwise_equal(ctx.read_radiance(), rgba_w)
        # configs[4] at its resolution: 7680x4320, one sample per pixel, <= 8 segments, all 33,177,600 pixels
        app.camera_set_target(7680, 4320)
        m8, p8 = app.camera_view()
        ctx.set_view(m8, p8, 7680, 4320)
        assert ctx.render(flags=rt.FLAG_PATH_TRACE, spp=1, max_segments=8, seed=2, stream_id=3) == rt.RT_OK
        acc8 = ctx.read_accumulation()
        acc8_w, segs8, _ = orc.render_path_tree(osc, tree, oracle_view(app), spp=1, max_segments=8, seed=2, stream=3, threads=16)
        c = ctx.counters()
        assert bitwise_equal(acc8, acc8_w) and c.rays_primary + c.rays_bounce == segs8
        del acc8, acc8_w
        app.camera_set_target(1920, 1080)
        # ... and the walk itself pinned by the reference-shaped loop on a pixel sample
        pix = np.random.default_rng(7).choice(1920 * 1080, 600, replace=False).astype(np.uint32)
        rgba_o, prim_o, t_o, _ = orc.render_primary(osc, view, orc.FLAG_SHADOW, threads=16, pixel_list=pix)
        assert bitwise_equal(prim_w.ravel()[pix], prim_o) and bitwise_equal(rgba_w.reshape(-1, 4)[pix], rgba_o)


def test_deep_tree_uses_the_private_part_of_the_stack():
    """600 k parallel triangles stacked along the view direction: the tree is deeper than the 16 stack levels a lane
    keeps in LDS, and a ray that runs down the stack pushes a far child on every level, so the levels beyond 16 (private
    array behind the LDS column) are exercised in every kernel variant."""
    n, W, H = 600_000, 32, 18
    z = -(np.arange(n, dtype=np.float32) * np.float32(0.01))
    verts = np.zeros(3 * n, rt.VERTEX)
    base = np.array([(-4.0, -3.0), (4.0, -3.0), (0.0, 5.0)], np.float32)      # counter-clockwise seen from +z: front face
    verts["pos"][:, :2] = np.tile(base, (n, 1))
    verts["pos"][:, 2] = np.repeat(z, 3)
    verts["norm"][:] = (0.0, 0.6, 0.8)
    tris = np.zeros(n, rt.TRIANGLE); tris["indices"] = np.arange(3 * n, dtype=np.uint32).reshape(n, 3)
    meshes = np.zeros(1, rt.MESH_INFO)
    meshes["first_tri"], meshes["tri_count"], meshes["material"] = 0, n, 0
    meshes["aabb_min"][0], meshes["aabb_max"][0] = (-5, -4, float(z[-1]) - 1), (5, 6, 1)
    mats = np.zeros(1, rt.MATERIAL); mats["color"][0] = (0.9, 0.5, 0.1, 1.0)
    scene = rt.Scene(tris, meshes, verts, mats)
    app = rt.App()
    app.camera_look_at((0.3, 0.2, 6.0), (0.0, 0.5, -50.0), (0, 1, 0))
    app.camera_set_target(W, H)
    m, p = app.camera_view()
    rgba_o, prim_o, t_o, _ = orc.render_primary(oracle_scene(scene), orc.make_view(m, p, W, H), orc.FLAG_SHADOW, threads=16)
    assert (prim_o == 0).sum() > 100                                         # the front triangle is what the rays see
    with rt.Context(0) as ctx:
        ctx.upload_scene(scene)
        assert ctx.bvh_info().max_depth > 16
        ctx.set_view(m, p, W, H)
        for extra in (0, rt.FLAG_STATIC_GRID, rt.FLAG_WAVE_TRAVERSAL):
            assert ctx.render(flags=rt.FLAG_SHADOW | extra) == rt.RT_OK
            assert bitwise_equal(ctx.read_prim_ids(), prim_o) and bitwise_equal(ctx.read_hit_t(), t_o)
            assert bitwise_equal(ctx.read_radiance(), rgba_o)
        # rays that look at the BACK of the stack hit nothing and have to walk the whole depth of the tree
        app.camera_look_at((0.3, 0.2, -6100.0), (0.0, 0.5, 0.0), (0, 1, 0))
        m2, p2 = app.camera_view()
        ctx.set_view(m2, p2, W, H)
        for extra in (0, rt.FLAG_STATIC_GRID):
            assert ctx.render(flags=extra | rt.FLAG_COUNTERS) == rt.RT_OK
            assert (ctx.read_prim_ids() == 0xFFFFFFFF).all()
            assert ctx.counters().tri_tests > 10_000                         # they walked into the stack, leaf after leaf


@pytest.mark.parametrize("list_max,levels_max", [("16384", "262144"), ("0", "0"), ("0", "262144")])
def test_update_vertex_range_random_sequences(list_max, levels_max, monkeypatch):
    """A sequence of random ranged updates (runs of 1-6 meshes, random moves, some back to where they were) on one context
    against full updates on another: after every step the binary tree, the records and the wide tree must be identical."""
    monkeypatch.setenv("RT_TUNE_REFIT_LIST_MAX", list_max)
    monkeypatch.setenv("RT_TUNE_REFIT_LEVELS_MAX", levels_max)
    app = soup_app(40, 0x5EED0007, 96, 54, subdivisions=2)
    app.camera_frame_scene(1.0)
    scene = app.prepare_meshinfo()
    cur = rt.Scene(scene.triangles, scene.meshes.copy(), scene.vertices.copy(), scene.materials)
    rng = np.random.default_rng(23)
    spans = []
    for mi in scene.meshes:
        vs = scene.triangles["indices"][mi["first_tri"]:mi["first_tri"] + mi["tri_count"]]
        spans.append((int(vs.min()), int(vs.max())))
    m, p = app.camera_view()
    with rt.Context(0) as full, rt.Context(0) as part:
        for ctx in (full, part):
            ctx.upload_scene(scene)
            ctx.set_view(m, p, 96, 54)
        for step in range(12):
            m0 = int(rng.integers(0, 40))
            m1 = min(40, m0 + int(rng.integers(1, 7)))
            for k in range(m0, m1):
                lo, hi = spans[k]
                if step % 4 == 3:                       # back to the uploaded pose
                    cur.vertices[lo:hi + 1] = scene.vertices[lo:hi + 1]
                    cur.meshes[k] = scene.meshes[k]
                else:
                    shift = rng.uniform(-3, 3, 3).astype(np.float32)
                    cur.vertices["pos"][lo:hi + 1] += shift
                    cur.meshes["aabb_min"][k] += shift
                    cur.meshes["aabb_max"][k] += shift
            v0, v1 = spans[m0][0], spans[m1 - 1][1]
            part.update_vertex_range(cur, v0, v1 - v0 + 1, m0, m1 - m0)
            full.update_vertices(cur)
            (nf, rf), (np_, rp) = full.read_bvh(), part.read_bvh()
            assert bitwise_equal(np_, nf) and bitwise_equal(rp, rf), step
            assert bitwise_equal(part.read_bvh_wide(), full.read_bvh_wide()), step
        part.render(flags=rt.FLAG_SHADOW)
        rgba_o, prim_o, t_o, _ = orc.render_primary(oracle_scene(cur), oracle_view(app), orc.FLAG_SHADOW)
        assert bitwise_equal(part.read_prim_ids(), prim_o) and bitwise_equal(part.read_radiance(), rgba_o)
