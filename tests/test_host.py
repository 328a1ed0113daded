"""CPU-side tests of the product's host logic (no GPU): the C-ABI library loads and exports every
declared symbol, the host mirror of prepare_meshinfo (src/ray_tracing.rs:98-169) agrees byte for byte
with the oracle's restatement incl. the reference's quirks, the mesh generators have the expected
counts, the BVH builder's flattened layout is a valid bounding hierarchy, and rt_create fails loudly
without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle_py as orc
import ray_tracing_amd as rt
from helpers import default_app, oracle_scene, soup_app

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    lib = rt.lib()
    declared = set()
    for h in ("rt_amd.h", "rt_host.h"):
        text = open(os.path.join(ROOT, "include", h)).read()
        declared |= set(re.findall(r"\b(rth?_[a-z0-9_]+)\s*\(", text))
    declared -= {"rt_ctx", "rth_app"}
    assert declared, "no declarations parsed"
    assert declared == set(rt.RT_SYMBOLS + rt.RTH_SYMBOLS), declared ^ set(rt.RT_SYMBOLS + rt.RTH_SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name
    assert b"gfx950" in lib.rt_version()


def test_layout_sizes_match_the_wgsl_bindings():
    assert (rt.TRIANGLE.itemsize, rt.MESH_INFO.itemsize, rt.VERTEX.itemsize, rt.MATERIAL.itemsize) == (12, 48, 32, 16)
    assert rt.MESH_INFO.fields["aabb_min"][1] == 16 and rt.MESH_INFO.fields["aabb_max"][1] == 32   # std430 vec3
    assert rt.VERTEX.fields["norm"][1] == 16
    assert C.sizeof(rt.RenderParams) == 48


def test_no_gpu_means_a_loud_failure_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(rt.RtError) as e:
        rt.Context(0)
    assert e.value.status == rt.RT_ERR_NO_GPU
    app = rt.App()
    app.spawn_default_scene()
    with pytest.raises(rt.RtError):
        app.finish(0)
    assert app.update() == rt.RT_NOT_READY      # pipeline not ready: frame skipped silently (src/node.rs:52-56)


def test_mesh_generators_have_the_expected_counts():
    app = rt.App()
    mat = app.add_material((1, 1, 1, 1))
    for mesh, nv, nt in ((app.add_cuboid(1, 2, 3), 24, 12), (app.add_plane(5, 5), 4, 2),
                         (app.add_icosphere(0.5, 5), 362, 720), (app.add_icosphere(2.0, 0), 12, 20),
                         (app.add_icosphere(1.0, 2), 92, 180)):
        app.clear_world()
        mat = app.add_material((1, 1, 1, 1))
    app = rt.App()
    mat = app.add_material((1, 1, 1, 1))
    app.spawn(app.add_cuboid(1, 2, 3), mat)
    app.spawn(app.add_plane(5, 5), mat)
    app.spawn(app.add_icosphere(0.5, 5), mat)
    app.spawn(app.add_icosphere(2.0, 0), mat)
    ents = app.entities()
    assert [(len(e["positions"]), len(e["indices"]) // 3) for e in ents] == [(24, 12), (4, 2), (362, 720), (12, 20)]
    sph = ents[2]
    assert np.allclose(np.linalg.norm(sph["positions"], axis=1), 0.5, atol=1e-6)
    assert np.allclose(sph["normals"] * 0.5, sph["positions"], atol=1e-6)
    # every triangle is wound counter-clockwise seen from outside (front faces for outside rays)
    for e in ents:
        p, ix = e["positions"], e["indices"].reshape(-1, 3)
        n = np.cross(p[ix[:, 1]] - p[ix[:, 0]], p[ix[:, 2]] - p[ix[:, 0]])
        c = (p[ix[:, 0]] + p[ix[:, 1]] + p[ix[:, 2]]) / 3
        if e is ents[1]:
            assert np.all(n[:, 1] > 0)                     # plane faces +Y
        else:
            assert np.all(np.einsum("ij,ij->i", n, c) > 0)


def test_default_scene_counts_and_materials():
    app = default_app()
    s = app.prepare_meshinfo()
    assert (len(s.triangles), len(s.meshes), len(s.vertices), len(s.materials)) == (2894, 6, 1476, 6)
    assert s.meshes["first_tri"].tolist() == [0, 12, 14, 734, 1454, 2174]
    assert s.meshes["material"].tolist() == [0, 1, 2, 3, 4, 5]            # one material per entity, no dedup
    assert s.materials["color"][0].tolist() == [0, 0, 1, 1] and s.materials["color"][2].tolist() == [1, 0, 0, 1]
    # Cuboid::from_size(half_extents*2).aabb_3d: the unit cube rotated 45 deg about X
    assert np.allclose(s.meshes["aabb_max"][0], [0.5, 0.70710677, 0.70710677], atol=1e-6)
    assert np.allclose(s.meshes["aabb_min"][2], [0.5, 0.0, 0.5]) and np.allclose(s.meshes["aabb_max"][2], [1.5, 1.0, 1.5])


def _same(a, b):
    return all(x.tobytes() == y.tobytes() for x, y in ((a.triangles, b.triangles), (a.meshes, b.meshes),
                                                       (a.vertices, b.vertices), (a.materials, b.materials)))


def test_prepare_meshinfo_matches_the_oracle_on_the_default_scene():
    app = default_app()
    assert _same(app.prepare_meshinfo(), orc.prepare_meshinfo(app.entities()))


def test_prepare_meshinfo_quirks_match_the_oracle():
    rng = np.random.default_rng(7)
    app = rt.App()
    mats = [app.add_material(rng.uniform(0, 1, 4)) for _ in range(5)]
    cube, sph = app.add_cuboid(1, 2, 0.5), app.add_icosphere(0.7, 1)
    # a mesh with no normals: its material is still pushed, no MeshInfo, index not bumped (:122-129)
    nonorm = app.add_mesh(rng.normal(size=(3, 3)), None, [0, 1, 2])
    # 7 indices: the trailing partial triple is dropped, tri_count = 7/3 = 2 (:152,161)
    odd = app.add_mesh(rng.normal(size=(5, 3)), rng.normal(size=(5, 3)), [0, 1, 2, 2, 3, 4, 0])
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    app.spawn(cube, mats[0], scale=(2, 0.5, 3), rotation=q, translation=(1, -2, 0.5))     # scale ignored by AABB/normals
    app.spawn(nonorm, mats[1])
    app.spawn(sph, mats[2], scale=(1, 1, 1), rotation=(0, 0, 0, 1), translation=(-3, 0, 2))
    e = app.spawn(odd, mats[3], rotation=q)
    app.set_aabb(e, (5, 5, 5), (0.25, 0.5, 0.75))                                          # centre ignored
    s = app.prepare_meshinfo()
    assert "Mesh missing attribute" in app.log()
    assert len(s.materials) == 4 and len(s.meshes) == 3
    assert s.meshes["material"].tolist() == [0, 1, 2]             # desynchronised from `materials` after the skip
    assert s.meshes["tri_count"].tolist() == [12, 80, 2]
    assert _same(s, orc.prepare_meshinfo(app.entities()))
    # normals: rotated, not scaled, not renormalised
    assert np.allclose(np.linalg.norm(s.vertices["norm"][:24], axis=1), 1.0, atol=1e-5)
    # AABB of the last mesh: |R| * half_extents about the translation (0,0,0), Aabb centre ignored
    assert np.allclose((s.meshes["aabb_min"][2] + s.meshes["aabb_max"][2]) / 2, 0, atol=1e-6)


def test_prepare_meshinfo_rebakes_only_what_moved_and_leaves_the_full_result():
    """f1, host side: with the same entities and assets, prepare_meshinfo re-bakes only the entities
    whose transform / Aabb changed, in place; what it leaves must be what a full pass (the oracle's) leaves, quirks kept."""
    rng = np.random.default_rng(3)
    app = rt.App()
    mats = [app.add_material(rng.uniform(0, 1, 4)) for _ in range(3)]
    cube, sph = app.add_cuboid(1, 2, 0.5), app.add_icosphere(0.7, 2)
    nonorm = app.add_mesh(rng.normal(size=(3, 3)), None, [0, 1, 2])
    ents = [app.spawn(cube, mats[0], translation=(1, 0, 0)), app.spawn(nonorm, mats[1]),
            app.spawn(sph, mats[2], translation=(-2, 0, 1)), app.spawn(sph, mats[0], translation=(3, 1, 1))]
    assert _same(app.prepare_meshinfo(), orc.prepare_meshinfo(app.entities()))
    for step in range(6):
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        k = ents[(0, 2, 3, 2, 0, 3)[step]]
        app.set_transform(k, scale=rng.uniform(0.5, 2, 3), rotation=q, translation=rng.uniform(-3, 3, 3))
        if step == 4:
            app.set_aabb(k, (0, 0, 0), (0.3, 0.2, 0.1))
        n_log = app.log().count("Mesh missing attribute")
        s = app.prepare_meshinfo()
        assert _same(s, orc.prepare_meshinfo(app.entities())), step
        assert app.log().count("Mesh missing attribute") == n_log + 1          # the skipped entity is reported every pass
    assert _same(app.prepare_meshinfo(), s)                                      # nothing moved: nothing changes
    # many entities at once (their runs are then re-baked on several threads)
    big = soup_app(60, 0x5EED0004, 64, 64)
    big.prepare_meshinfo()
    for k, e in enumerate(big.entities()):
        a = 0.1 * (k + 1)
        big.set_transform(k, e["scale"], (0.0, float(np.sin(a)), 0.0, float(np.cos(a))), e["translation"] + np.float32(0.25))
    assert _same(big.prepare_meshinfo(), orc.prepare_meshinfo(big.entities()))
    # a new asset, or a world cleared and refilled with as many entities: a full pass again
    app.spawn(app.add_cuboid(2, 2, 2), mats[1], translation=(0, 5, 0))
    assert _same(app.prepare_meshinfo(), orc.prepare_meshinfo(app.entities()))
    app.clear_world()
    m = app.add_material((1, 1, 1, 1))
    for k in range(5):
        app.spawn(app.add_icosphere(0.5 + 0.1 * k, 1), m, translation=(k, 0, 0))
    assert _same(app.prepare_meshinfo(), orc.prepare_meshinfo(app.entities()))


def test_unindexed_mesh_is_the_references_panic():
    app = rt.App()
    m = app.add_material((1, 1, 1, 1))
    app.spawn(app.add_mesh(np.zeros((3, 3)), np.zeros((3, 3)), None), m)
    with pytest.raises(RuntimeError, match="no indices"):
        app.prepare_meshinfo()


def test_view_uniform_for_the_default_camera():
    app = default_app(640, 360)
    m, p = app.camera_view()
    assert np.array_equal(p, np.float32([0, 3, 5]))
    M = m.reshape(4, 4).T.astype(np.float64)
    # clip (0,0,0,1) -> direction at infinity along the camera's forward axis (towards the origin)
    fwd = M @ np.array([0, 0, 0, 1.0])
    assert abs(fwd[3]) < 1e-12 and np.allclose(fwd[:3] / np.linalg.norm(fwd[:3]), -p / np.linalg.norm(p), atol=1e-6)
    # clip (x,y,1,1): a finite point on the near plane (reverse-Z: z=1 is near = 0.1)
    near = M @ np.array([0, 0, 1.0, 1.0])
    near = near[:3] / near[3]
    assert abs(np.linalg.norm(near - p) - 0.1) < 1e-5
    # vertical extent of the frustum: tan(fov/2) = tan(pi/8)
    top = M @ np.array([0, 1.0, 0, 1.0])
    cosang = np.dot(top[:3], fwd[:3]) / np.linalg.norm(top[:3]) / np.linalg.norm(fwd[:3])
    assert abs(np.arccos(cosang) - np.pi / 8) < 1e-5


def test_fly_cam_restatement():
    app = default_app(1280, 720)
    m0, p0 = app.camera_view()
    app.fly_cam_step(keys=1, dt=0.5)                      # forward: planar, speed 12 (src/fly_cam.rs:108-136)
    m1, p1 = app.camera_view()
    assert abs(p1[1] - p0[1]) < 1e-6 and np.allclose(np.linalg.norm(p1 - p0), 6.0, atol=1e-4)
    assert p1[2] < p0[2]
    app.fly_cam_step(mouse_dx=100.0)                      # yaw -= 0.00012*100*720 degrees
    m2, _ = app.camera_view()
    f1, f2 = m1[12:15], m2[12:15]
    ang = np.degrees(np.arccos(np.clip(np.dot([f1[0], f1[2]], [f2[0], f2[2]]) / np.linalg.norm([f1[0], f1[2]]) / np.linalg.norm([f2[0], f2[2]]), -1, 1)))
    assert abs(ang - 8.64) < 0.01
    for _ in range(100):
        app.fly_cam_step(mouse_dy=-10000.0)               # pitch clamps at +-1.54 rad (:166)
    m3, _ = app.camera_view()
    assert abs(np.arcsin(np.clip(m3[13], -1, 1)) - 1.54) < 1e-3


def test_fly_cam_script_is_the_same_steps_from_text():
    """the fly-cam driven from a scripted input file instead of window events."""
    script = """# dt   mouse_dx mouse_dy keys
    0.5   0      0       W          # forward for half a second
    0.0   100    0       -          # look left
    0.25  -30    12      WD^        # forward + right + ascend while looking around
    0.1   0      0                  # no key column: nothing pressed
    0.2   5      -7      sav        # lower case, backward + left + descend
    """
    a, b = default_app(1280, 720), default_app(1280, 720)
    assert a.fly_cam_run_script(script) == 5
    K = {"W": 1, "S": 2, "A": 4, "D": 8, "^": 16, "v": 32}
    for dt, mx, my, keys in ((0.5, 0, 0, "W"), (0.0, 100, 0, ""), (0.25, -30, 12, "WD^"), (0.1, 0, 0, ""), (0.2, 5, -7, "SAv")):
        b.fly_cam_step(mouse_dx=mx, mouse_dy=my, keys=sum(K[k] for k in keys), dt=dt)
    ma, pa = a.camera_view()
    mb, pb = b.camera_view()
    assert ma.tobytes() == mb.tobytes() and pa.tobytes() == pb.tobytes()
    m0, p0 = default_app(1280, 720).camera_view()
    assert pa.tobytes() != p0.tobytes()
    with pytest.raises(ValueError, match="line 2"):
        default_app().fly_cam_run_script("0.1 0 0 W\n0.1 0 Q\n")
    with pytest.raises(ValueError, match="line 1"):
        default_app().fly_cam_run_script("0.1 0 0 X")


def test_image_writers_round_trip(tmp_path):
    """the presented BGRA8 sRGB frame as PPM and PNG; both read back to the same pixels."""
    import struct
    import zlib
    rng = np.random.default_rng(1)
    h, w = 37, 53                       # ragged sizes; > 65535 raw bytes would need two stored blocks: see below
    bgra = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)
    rt.write_image(tmp_path / "f.ppm", bgra)
    raw = (tmp_path / "f.ppm").read_bytes()
    head = b"P6\n53 37\n255\n"
    assert raw.startswith(head) and len(raw) == len(head) + h * w * 3
    rgb = np.frombuffer(raw[len(head):], np.uint8).reshape(h, w, 3)
    assert np.array_equal(rgb, bgra[..., [2, 1, 0]])

    def read_png(path):
        data = path.read_bytes()
        assert data[:8] == b"\x89PNG\r\n\x1a\n"
        at, chunks = 8, []
        while at < len(data):
            n, typ = struct.unpack(">I4s", data[at:at + 8])
            body = data[at + 8:at + 8 + n]
            assert zlib.crc32(typ + body) == struct.unpack(">I", data[at + 8 + n:at + 12 + n])[0]
            chunks.append((typ, body))
            at += 12 + n
        assert [c[0] for c in chunks] == [b"IHDR", b"sRGB", b"IDAT", b"IEND"]
        W, H, depth, ctype, comp, filt, inter = struct.unpack(">IIBBBBB", chunks[0][1])
        assert (depth, ctype, comp, filt, inter) == (8, 6, 0, 0, 0)
        rows = np.frombuffer(zlib.decompress(chunks[2][1]), np.uint8).reshape(H, W * 4 + 1)
        assert np.all(rows[:, 0] == 0)
        return rows[:, 1:].reshape(H, W, 4)
    rt.write_image(tmp_path / "f.png", bgra)
    assert np.array_equal(read_png(tmp_path / "f.png"), bgra[..., [2, 1, 0, 3]])
    big = rng.integers(0, 256, size=(200, 300, 4), dtype=np.uint8)      # 240,200 raw bytes: four stored blocks
    rt.write_image(tmp_path / "big.png", big)
    assert np.array_equal(read_png(tmp_path / "big.png"), big[..., [2, 1, 0, 3]])
    with pytest.raises(OSError):
        rt.write_image(tmp_path / "no_such_dir" / "f.png", bgra)


def _check_bvh(scene, nodes, recs, info):
    n_items = int(scene.meshes["tri_count"].sum())
    assert len(recs) == n_items and sorted(recs["seq"].tolist()) == list(range(n_items))
    V = scene.vertices["pos"]
    T = scene.triangles["indices"]
    a, b, c = V[T[recs["tri"], 0]], V[T[recs["tri"], 1]], V[T[recs["tri"], 2]]
    assert np.array_equal(recs["a"], a) and np.array_equal(recs["ab"], b - a) and np.array_equal(recs["ac"], c - a)
    first = scene.meshes["first_tri"][recs["mesh"]]
    assert np.all((recs["tri"] >= first) & (recs["tri"] < first + scene.meshes["tri_count"][recs["mesh"]]))
    lo_t, hi_t = np.minimum(np.minimum(a, b), c), np.maximum(np.maximum(a, b), c)
    seen = np.zeros(len(recs), bool)
    visited = np.zeros(len(nodes), bool)
    max_depth = 0

    def walk(code, lo, hi, depth):
        nonlocal max_depth
        max_depth = max(max_depth, depth)
        if code >= 0:
            assert not visited[code]
            visited[code] = True
            n = nodes[code]
            for ch, l, h in ((n["child0"], n["lo0"], n["hi0"]), (n["child1"], n["lo1"], n["hi1"])):
                if ch == -2**31:
                    assert np.all(np.isnan(l))
                    continue
                assert np.all(l >= lo - 1e-6) and np.all(h <= hi + 1e-6)
                walk(int(ch), l, h, depth + 1)
        else:
            v = ~code
            f, k = v >> 3, (v & 7) + 1
            assert k <= info.max_leaf <= 4
            assert not seen[f:f + k].any()
            seen[f:f + k] = True
            assert np.all(lo_t[f:f + k] >= lo) and np.all(hi_t[f:f + k] <= hi)      # padded boxes contain the triangles

    inf = np.full(3, np.inf, np.float32)
    if len(nodes):
        walk(0, -inf, inf, 0)
    assert seen.all() and visited.all() and max_depth <= info.max_depth + 1 < 48


def test_bvh_builder_layout_is_a_valid_hierarchy():
    import sys
    sys.setrecursionlimit(10000)
    for app in (default_app(), soup_app(12, 3, 64, 64, subdivisions=2)):
        scene = app.prepare_meshinfo()
        info, nodes, recs = rt.host_bvh_build(scene, threads=2)
        assert info.node_bytes == 64 and info.tri_bytes == 48
        _check_bvh(scene, nodes, recs, info)


def test_bvh_refit_keeps_a_valid_hierarchy_over_moved_vertices():
    import sys
    sys.setrecursionlimit(10000)
    app = default_app()
    scene = app.prepare_meshinfo()
    app.rotate_tagged(0.7)                       # the `rotate` system moves the cube (src/main.rs:122-126)
    moved = app.prepare_meshinfo()
    assert moved.triangles.tobytes() == scene.triangles.tobytes() and moved.vertices.tobytes() != scene.vertices.tobytes()
    info, nodes, recs = rt.host_bvh_build(scene, threads=2, refit_to=moved)
    _check_bvh(moved, nodes, recs, info)         # boxes bound the MOVED triangles, records re-baked
    _, nodes0, recs0 = rt.host_bvh_build(scene, threads=2)
    assert np.array_equal(nodes["child0"], nodes0["child0"]) and np.array_equal(recs["seq"], recs0["seq"])   # topology kept
    bad = rt.Scene(moved.triangles, moved.meshes, moved.vertices.copy(), moved.materials)
    bad.vertices["pos"][3] = (np.inf, 0, 0)
    with pytest.raises(rt.RtError):
        rt.host_bvh_build(scene, refit_to=bad)   # a non-finite vertex needs a rebuild


def test_bvh_builder_edge_cases():
    empty = rt.Scene(np.zeros(0, rt.TRIANGLE), np.zeros(0, rt.MESH_INFO), np.zeros(0, rt.VERTEX), np.zeros(0, rt.MATERIAL))
    info, nodes, recs = rt.host_bvh_build(empty)
    assert info.n_nodes == 0 and info.n_triangles == 0
    # one triangle: a root whose second child is absent (NaN box)
    v = np.zeros(3, rt.VERTEX)
    v["pos"] = [(0, 0, 0), (0, 0, 1), (1, 0, 0)]
    t = np.zeros(1, rt.TRIANGLE)
    t["indices"] = (0, 1, 2)
    m = np.zeros(1, rt.MESH_INFO)
    m["tri_count"] = 1
    one = rt.Scene(t, m, v, np.zeros(1, rt.MATERIAL))
    info, nodes, recs = rt.host_bvh_build(one)
    assert info.n_nodes == 1 and nodes["child1"][0] == -2**31 and np.isnan(nodes["lo1"][0]).all()
    # many coincident triangles (identical centroids): falls back to median splits, still a valid tree
    t = np.zeros(100, rt.TRIANGLE)
    t["indices"] = (0, 1, 2)
    m["tri_count"] = 100
    many = rt.Scene(t, m, v, np.zeros(1, rt.MATERIAL))
    info, nodes, recs = rt.host_bvh_build(many)
    _check_bvh(many, nodes, recs, info)
    # a NaN vertex drops its triangle from the tree (it can never be hit)
    v2 = v.copy()
    v2["pos"][1] = (np.nan, 0, 0)
    m1 = m.copy()
    m1["tri_count"] = 1
    info, nodes, recs = rt.host_bvh_build(rt.Scene(t[:1], m1, v2, np.zeros(1, rt.MATERIAL)))
    assert info.n_triangles == 0
    # out-of-range index / mesh range are refused before anything is built
    bad = t[:1].copy()
    bad["indices"] = (0, 1, 9)
    with pytest.raises(rt.RtError) as e:
        rt.host_bvh_build(rt.Scene(bad, m1, v, np.zeros(1, rt.MATERIAL)))
    assert e.value.status == rt.RT_ERR_SCENE
    m2 = m.copy()
    m2["tri_count"] = 5
    with pytest.raises(rt.RtError):
        rt.host_bvh_build(rt.Scene(t[:1], m2, v, np.zeros(1, rt.MATERIAL)))


def test_cpu_walk_of_the_exported_bvh_finds_the_oracles_hits():
    """A plain-Python walk of the product's flattened BVH (closest hit by lowest t then lowest seq,
    triangle test by the oracle) gives the oracle's brute-force answer: validates builder + layout."""
    app = default_app(24, 14)
    scene = app.prepare_meshinfo()
    osc = oracle_scene(scene)
    info, nodes, recs = rt.host_bvh_build(scene)
    m, p = app.camera_view()
    view = orc.make_view(m, p, 24, 14)
    _, prim_o, t_o, _ = orc.render_primary(osc, view, orc.FLAG_NO_MESH_CULL)
    for py in range(0, 14, 2):
        for px in range(0, 24, 3):
            o, d = orc.ray_gen(view, px, py)
            with np.errstate(divide="ignore", invalid="ignore"):
                inv = np.float32(1.0) / d
            best = (np.inf, 2**32, 0xFFFFFFFF)
            stack = [0]
            while stack:
                code = stack.pop()
                if code >= 0:
                    n = nodes[code]
                    for ch, lo, hi in ((n["child0"], n["lo0"], n["hi0"]), (n["child1"], n["lo1"], n["hi1"])):
                        if ch == -2**31:
                            continue
                        with np.errstate(invalid="ignore"):
                            t1, t2 = (lo - o) * inv, (hi - o) * inv
                            tmin, tmax = np.fmax.reduce(np.fmin(t1, t2)), np.fmin.reduce(np.fmax(t1, t2))
                        if tmin <= tmax * 1.0000005 and tmax >= 0 and tmin <= best[0] * 1.00001:
                            stack.append(int(ch))
                else:
                    v = ~code
                    for k in range(v >> 3, (v >> 3) + (v & 7) + 1):
                        h = orc.ray_triangle(osc, o, d, int(recs["tri"][k]))
                        if h["hit"] and (h["t"], int(recs["seq"][k])) < best[:2]:
                            best = (h["t"], int(recs["seq"][k]), int(recs["tri"][k]))
            assert best[2] == prim_o[py, px]
            assert (best[0] if best[2] != 0xFFFFFFFF else 0.0) == t_o[py, px]


def test_bvh_is_the_same_tree_whatever_the_thread_count():
    """The builder's threads take whole subtrees and, in nodes of >= 2^18 primitives, shares of the passes over the
    primitives (bounds, binning): minima, maxima and counts do not depend on how a run is cut up, so binary tree, leaf
    records and wide tree must be identical, byte for byte, for any thread count."""
    scene = soup_app(420, 0x5EED0009, 64, 64).prepare_meshinfo()          # 302,400 triangles: the root takes the parallel passes
    assert len(scene.triangles) > (1 << 18)
    ref = None
    for threads in (1, 3, 8):
        info, nodes, recs, wn, ws, need = rt.host_bvh_build(scene, threads=threads, wide=True)
        cur = (nodes.tobytes(), recs.tobytes(), wn.tobytes(), ws.tobytes(), need, info.max_depth, info.max_leaf)
        if ref is None:
            ref = cur
        assert cur == ref, threads


def test_rust_sys_source_binds_every_entry_point_of_the_header():
    """integration/rt-amd-sys/src/lib.rs is unbuilt source (no Rust toolchain in the image): at least its list of `pub fn`
    must be the header's list of entry points, so the binding a maintainer would compile cannot silently fall behind."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "rt_amd.h")).read()
    rust = open(os.path.join(root, "integration", "rt-amd-sys", "src", "lib.rs")).read()
    declared = set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", header))
    bound = set(re.findall(r"pub fn (rt_[a-z0-9_]+)", rust))
    assert declared == bound, (sorted(declared - bound), sorted(bound - declared))
    assert declared == set(rt.RT_SYMBOLS)


def test_rust_node_shim_source_is_complete_against_the_sys_crate_and_itself():
    """integration/node_shim.rs is unbuilt source too (it called three functions that nowhere existed).  What can be
    held without a compiler: every rt_* function, type and constant it names is declared by the -sys crate; every free
    function it calls is defined in the file; it uses the round-2 API the header recommends for a node that keeps submitting
    (dirty-range updates, frames in flight, asynchronous presentation) and not the blocking read."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    shim = open(os.path.join(root, "integration", "node_shim.rs")).read()
    code = re.sub(r'r#".*?"#', '""', shim, flags=re.S)                                  # the WGSL text is not Rust
    code = "\n".join(line.split("//")[0] for line in code.splitlines() if not line.lstrip().startswith("//"))
    names_in_strings = set(re.findall(r"\b(rt_[a-z0-9_]+)\b", " ".join(re.findall(r'"((?:[^"\\]|\\.)*)"', code))))
    code = re.sub(r'"(?:[^"\\]|\\.)*"', '""', code)                                   # messages and labels neither
    rust = open(os.path.join(root, "integration", "rt-amd-sys", "src", "lib.rs")).read()
    sys_names = set(re.findall(r"pub fn (rt_[a-z0-9_]+)", rust)) | set(re.findall(r"pub struct (rt_[a-z0-9_]+)", rust)) | \
        set(re.findall(r"pub const (RT_[A-Z0-9_]+)", rust))
    used = set(re.findall(r"\b(rt_[a-z0-9_]+)\b", code)) - {"rt_amd_sys"}
    assert names_in_strings                                                               # (error messages name the calls they report)
    used |= set(re.findall(r"\b(RT_[A-Z0-9_]+)\b", code))
    assert used <= sys_names, sorted(used - sys_names)
    defined = set(re.findall(r"\bfn ([a-z_0-9]+)\s*[<(]", code))
    called = set(re.findall(r"(?<![.:\w])([a-z_][a-z_0-9]*)\s*\(", code)) - sys_names
    rust_builtins = {"map", "matches", "panic", "assert_eq", "vec", "single", "filter", "if", "match", "for", "fn", "unsafe", "return", "texture_2d",
                     "fullscreen_shader_vertex_state", "derive", "load", "let"}
    missing = {c for c in called - defined - rust_builtins if not c[0].isupper()}
    assert not missing, sorted(missing)
    for name in ("to_repr_c", "signatures", "write_to_target"):          # the three that did not exist
        assert name in defined
    for api in ("rt_update_vertex_range", "rt_set_frames_in_flight", "rt_present_async", "rt_present_wait"):
        assert api in used
    assert "rt_read_bgra8_srgb" not in used
