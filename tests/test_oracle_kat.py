"""Pins of the CPU oracle (oracle/rt_oracle.c): the hand-worked known-answer tests KAT-1..6 of
SURVEY.md section 4 (derived directly from assets/shaders/ray_tracing.wgsl:118-170,203-211), RNG
known answers from wgsl:39-43 restated in pure Python, and the committed golden fixtures.
The reference itself has no tests or vectors ("parity unpinned"): these are the independent pins."""
import os

import numpy as np
import pytest

import oracle_py as orc

HERE = os.path.dirname(os.path.abspath(__file__))


def one_triangle_scene(normals=None, second_mesh=False):
    # KAT triangle: A=(0,0,0), B=(0,0,1), C=(1,0,0); cross(AB, AC) = (0,1,0)
    v = np.zeros(6 if second_mesh else 3, orc.VERTEX)
    pos = [(0, 0, 0), (0, 0, 1), (1, 0, 0)]
    nrm = normals or [(0, 1, 0)] * 3
    for i in range(3):
        v[i]["pos"], v[i]["norm"] = pos[i], nrm[i]
        if second_mesh:
            v[i + 3]["pos"], v[i + 3]["norm"] = pos[i], nrm[i]
    t = np.zeros(2 if second_mesh else 1, orc.TRIANGLE)
    t[0]["indices"] = (0, 1, 2)
    m = np.zeros(2 if second_mesh else 1, orc.MESH_INFO)
    m[0]["first_tri"], m[0]["tri_count"], m[0]["material"] = 0, 1, 0
    m[0]["aabb_min"], m[0]["aabb_max"] = (-1, -1, -1), (2, 1, 2)
    c = np.zeros(2, orc.MATERIAL)
    c[0]["color"] = (0.25, 0.5, 1.0, 1.0)
    c[1]["color"] = (1.0, 0.0, 0.0, 1.0)
    if second_mesh:
        t[1]["indices"] = (3, 4, 5)
        m[1]["first_tri"], m[1]["tri_count"], m[1]["material"] = 1, 1, 1
        m[1]["aabb_min"], m[1]["aabb_max"] = (-1, -1, -1), (2, 1, 2)
    return orc.Scene(t, m, v, c)


def test_kat1_front_hit():
    s = one_triangle_scene()
    h = orc.hit_triangles(s, (0.25, 1, 0.25), (0, -1, 0))
    assert h["hit"] and h["t"] == 1.0 and h["prim"] == 0 and h["material"] == 0
    assert np.array_equal(h["point"], np.float32([0.25, 0, 0.25]))
    assert np.array_equal(h["normal"], np.float32([0, 1, 0]))
    assert np.array_equal(orc.shade(s, h), np.float32([0.25, 0.5, 1.0, 1.0]))   # colour x dot(n,+Y)=1


def test_kat2_back_face_is_culled():
    s = one_triangle_scene()
    h = orc.hit_triangles(s, (0.25, -1, 0.25), (0, 1, 0))     # det = -1 < 1e-6
    assert not h["hit"]
    r = orc.ray_triangle(s, (0.25, -1, 0.25), (0, 1, 0), 0)
    assert not r["hit"]


def test_kat3_tie_goes_to_the_earlier_triangle():
    s = one_triangle_scene(second_mesh=True)                 # two coincident triangles, two meshes
    h = orc.hit_triangles(s, (0.25, 1, 0.25), (0, -1, 0))
    assert h["hit"] and h["prim"] == 0 and h["material"] == 0  # strict `hit.t > record.t` (wgsl:164)


def test_kat4_aabb_is_a_cull_only():
    # box fully behind the ray still passes: no tmax >= 0 test (wgsl:147)
    assert orc.ray_aabb((0, 0, 0), (0, 0, 1), (-1, -1, -5), (1, 1, -3))
    assert not orc.ray_aabb((0, 0, 0), (0, 0, 1), (2, 2, -5), (3, 3, -3))
    # correctness rests on dst >= 0: a triangle behind the origin is not hit
    s = one_triangle_scene()
    assert not orc.hit_triangles(s, (0.25, -1, 0.25), (0, -1, 0))["hit"]


def test_kat5_u_weights_b_and_v_weights_c():
    s = one_triangle_scene(normals=[(1, 0, 0), (0, 1, 0), (0, 0, 1)])
    # hit point (0.25, 0, 0.5): P = A + u*AB + v*AC with AB=(0,0,1), AC=(1,0,0) -> u=0.5, v=0.25, w=0.25
    h = orc.hit_triangles(s, (0.25, 1, 0.5), (0, -1, 0))
    n = np.float32([0.25, 0.5, 0.25])        # a.n*w + b.n*u + c.n*v
    n = n / np.float32(np.sqrt(np.float32(np.dot(n, n))))
    assert h["hit"] and np.allclose(h["normal"], n, rtol=0, atol=1e-7)


def test_kat6_miss_colour_is_zero_including_alpha():
    s = one_triangle_scene()
    h = orc.hit_triangles(s, (5, 1, 5), (0, -1, 0))
    assert not h["hit"] and np.array_equal(orc.shade(s, h), np.zeros(4, np.float32))


def test_mesh_aabb_cull_hides_triangles_outside_their_mesh_box():
    s = one_triangle_scene()
    s.meshes[0]["aabb_min"], s.meshes[0]["aabb_max"] = (10, 10, 10), (11, 11, 11)   # wrong box (P1 quirk)
    assert not orc.hit_triangles(s, (0.25, 1, 0.25), (0, -1, 0))["hit"]
    assert orc.hit_triangles(s, (0.25, 1, 0.25), (0, -1, 0), orc.FLAG_NO_MESH_CULL)["hit"]


def test_debug_grey_and_out_of_range_material():
    s = one_triangle_scene()
    s.meshes[0]["material"] = 103
    h = orc.hit_triangles(s, (0.25, 1, 0.25), (0, -1, 0))
    assert np.allclose(orc.shade(s, h), np.float32(0.5 + 3 / 6.0))            # wgsl:205-207
    s.meshes[0]["material"] = 7                                               # >= n_mat, < 100
    h = orc.hit_triangles(s, (0.25, 1, 0.25), (0, -1, 0))
    assert np.array_equal(orc.shade(s, h), np.zeros(4, np.float32))


def pcg_next(state):          # ray_tracing.wgsl:39-43 in pure Python integers
    state = (state * 747796405 + 2891336453) & 0xFFFFFFFF
    word = (((state >> ((state >> 28) + 4)) ^ state) * 277803737) & 0xFFFFFFFF
    return state, ((word >> 22) ^ word) & 0xFFFFFFFF


def test_rng_integer_core_matches_the_wgsl_formula():
    import ctypes as C
    for seed in (1, 0, 12345, 0xFFFFFFFF, 0x9E3779B9):
        st = C.c_uint32(seed)
        ref = seed
        for _ in range(16):
            ref, out = pcg_next(ref)
            assert orc.lib().orc_next_random(C.byref(st)) == out and st.value == ref
    st = C.c_uint32(1)
    x = orc.lib().orc_rand(C.byref(st))
    _, w = pcg_next(1)
    assert x == np.float32(np.float32(w) / np.float32(4294967295.0))         # divisor rounds to 2^32


def test_deterministic_log_and_cos_are_accurate():
    xs = np.concatenate([np.linspace(1e-6, 1.0, 4001), np.float32([1e-30, 1e-38, 3.0, 100.0, 1e20])]).astype(np.float32)
    got = np.float32([orc.lib().orc_det_logf(float(x)) for x in xs])
    assert np.max(np.abs(got - np.log(xs.astype(np.float64))) / np.maximum(1.0, np.abs(np.log(xs.astype(np.float64))))) < 2e-7
    th = np.linspace(0.0, 2 * 3.1415926, 5001).astype(np.float32)
    got = np.float32([orc.lib().orc_det_cosf(float(x)) for x in th])
    assert np.max(np.abs(got - np.cos(th.astype(np.float64)))) < 3e-7
    assert np.isneginf(orc.lib().orc_det_logf(0.0)) and np.isnan(orc.lib().orc_det_logf(-1.0))


def test_rand_hemi_is_unit_and_in_the_normal_hemisphere():
    import ctypes as C
    st = C.c_uint32(99)
    n = (C.c_float * 3)(0.0, 0.6, 0.8)
    d = (C.c_float * 3)()
    for _ in range(200):
        orc.lib().orc_rand_hemi(C.byref(st), n, d)
        v = np.float32(list(d))
        assert abs(float(np.linalg.norm(v)) - 1.0) < 1e-5 and float(np.dot(v, [0.0, 0.6, 0.8])) >= 0.0


def test_ray_gen_matches_the_wgsl_expression_in_float64():
    rng = np.random.default_rng(1)
    m = rng.normal(size=16).astype(np.float32)
    p = rng.normal(size=3).astype(np.float32)
    v = orc.make_view(m, p, 40, 20)
    for px, py in ((0, 0), (39, 19), (17, 3)):
        o, d = orc.ray_gen(v, px, py)
        uv = np.array([(px + 0.5) / 40 * 2 - 1, (py + 0.5) / 20 * -2 + 1, 0.0, 1.0])
        r = m.astype(np.float64).reshape(4, 4).T @ uv                  # column-major
        r = r / np.linalg.norm(r)                                      # normalize over the vec4
        assert np.array_equal(o, p) and np.allclose(d, r[:3], rtol=0, atol=2e-6)


def test_viewport_restricts_and_rebases_uv():
    m = np.eye(4, dtype=np.float32).ravel()
    full = orc.make_view(m, (0, 0, 0), 32, 16)
    vp = orc.make_view(m, (0, 0, 0), 64, 64, (8, 4, 32, 16))
    for px, py in ((0, 0), (31, 15), (5, 9)):
        assert np.array_equal(orc.ray_gen(full, px, py)[1], orc.ray_gen(vp, px + 8, py + 4)[1])


def test_srgb_bgra8_encode_known_values():
    rgba = np.float32([[0.0, 0.5, 1.0, 1.0], [-1.0, 2.0, 0.0031308, 0.5], [np.nan, 0.2, 0.8, 2.0]])
    out = orc.encode_bgra8_srgb(rgba)
    assert out[0].tolist() == [255, 188, 0, 255]            # B, G, R, A ; srgb(0.5) = 187.5 -> 188
    assert out[1].tolist() == [10, 255, 0, 128]             # 0.0031308*12.92*255 = 10.3
    assert out[2][2] == 0 and out[2][3] == 255              # NaN -> 0, alpha clamps


@pytest.mark.parametrize("name", ["default_64x36_shadow", "spheres8_48x32"])
def test_oracle_reproduces_the_golden_fixtures(name):
    g = np.load(os.path.join(HERE, "golden", name + ".npz"))
    s = orc.Scene(g["triangles"], g["meshes"], g["vertices"], g["materials"])
    W, H = (int(x) for x in g["size"])
    v = orc.make_view(g["world_from_clip"], g["world_position"], W, H)
    rgba, prim, t, rays = orc.render_primary(s, v, int(g["flags"]), threads=3)
    assert rgba.tobytes() == g["rgba"].tobytes() and prim.tobytes() == g["prim"].tobytes()
    assert t.tobytes() == g["t"].tobytes() and rays == int(g["rays"])
    if "path_accum" in g:
        acc, segs = orc.render_path(s, v, int(g["path_spp"]), 4, seed=int(g["path_seed"]), threads=5)
        assert acc.tobytes() == g["path_accum"].tobytes() and segs == int(g["path_segments"])


def test_path_accumulation_is_independent_of_frame_slicing_and_threads():
    g = np.load(os.path.join(HERE, "golden", "spheres8_48x32.npz"))
    s = orc.Scene(g["triangles"], g["meshes"], g["vertices"], g["materials"])
    v = orc.make_view(g["world_from_clip"], g["world_position"], 48, 32)
    a, _ = orc.render_path(s, v, 4, 3, seed=5, threads=1)
    b, _ = orc.render_path(s, v, 3, 3, seed=5, threads=7)
    b, _ = orc.render_path(s, v, 1, 3, seed=5, sample_first=3, accum=b, threads=2)
    assert a.tobytes() == b.tobytes()
    assert np.all(a[..., 3] == 4.0)
    assert np.all(orc.resolve(a, 4)[..., :3] <= 0.86 + 1e-6)         # never brighter than the sky
