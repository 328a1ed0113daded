"""A SECOND, independent restatement of the live path -- numpy float32, written from the WGSL text alone
(assets/shaders/ray_tracing.wgsl:118-170, 173-177, 203-211), vectorised over rays -- against the C oracle and the committed
golden frames, bit for bit.  The reference holds no fixtures and cannot run here, so parity stays "unpinned" by
it; what this test removes is the risk that ONE transcription (oracle/rt_oracle.c) carries a slip that the GPU kernels,
written against that oracle, would faithfully reproduce.  Float contract as in the oracle's header: binary32 throughout,
dot = (x*x' + y*y') + z*z', cross component = p*q - r*s, normalize = v / sqrt(dot(v, v)), min/max drop a NaN operand."""
import os

import numpy as np
import pytest

import oracle_py as orc

HERE = os.path.dirname(os.path.abspath(__file__))
f32 = np.float32


def dot(a, b):
    return (a[..., 0] * b[..., 0] + a[..., 1] * b[..., 1]) + a[..., 2] * b[..., 2]


def cross(a, b):
    return np.stack([a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1],
                     a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2],
                     a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]], axis=-1)


def normalize(a):
    return a / np.sqrt(dot(a, a))[..., None]


def ray_gen(m, pos, W, H):
    """wgsl:174,176-177: uv of the fullscreen triangle at pixel centres; normalize over the vec4, then .xyz."""
    ys, xs = np.mgrid[0:H, 0:W]
    u = (xs.astype(f32) + f32(0.5)) / f32(W)
    w = (ys.astype(f32) + f32(0.5)) / f32(H)
    nx = u * f32(2.0) + f32(-1.0)
    ny = w * f32(-2.0) + f32(1.0)
    r = [((m[0 + i] * nx + m[4 + i] * ny) + m[8 + i] * f32(0.0)) + m[12 + i] * f32(1.0) for i in range(4)]
    ln = np.sqrt(((r[0] * r[0] + r[1] * r[1]) + r[2] * r[2]) + r[3] * r[3])
    d = np.stack([r[0] / ln, r[1] / ln, r[2] / ln], axis=-1)
    o = np.broadcast_to(pos.astype(f32), d.shape).copy()
    return o.reshape(-1, 3), d.reshape(-1, 3)


def ray_aabb(o, d, lb, rt):
    """wgsl:141-148"""
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = f32(1.0) / d
        t1 = (lb - o) * inv
        t2 = (rt - o) * inv
    tmin = np.fmax(np.fmax(np.fmin(t1[:, 0], t2[:, 0]), np.fmin(t1[:, 1], t2[:, 1])), np.fmin(t1[:, 2], t2[:, 2]))
    tmax = np.fmin(np.fmin(np.fmax(t1[:, 0], t2[:, 0]), np.fmax(t1[:, 1], t2[:, 1])), np.fmax(t1[:, 2], t2[:, 2]))
    return tmax >= tmin


def hit_triangles(o, d, tris, meshes, verts):
    """wgsl:150-170 with ray_triangle (wgsl:118-139) inlined; returns hit, t, prim, material, point, normal."""
    n = o.shape[0]
    hit = np.zeros(n, bool)
    t = np.zeros(n, f32)
    prim = np.full(n, 0xFFFFFFFF, np.uint32)
    mat = np.zeros(n, np.uint32)
    point = np.zeros((n, 3), f32)
    normal = np.zeros((n, 3), f32)
    pos, nrm = verts["pos"].astype(f32), verts["norm"].astype(f32)
    for mesh in meshes:
        inside = ray_aabb(o, d, mesh["aabb_min"].astype(f32), mesh["aabb_max"].astype(f32))
        if not inside.any():
            continue
        for j in range(int(mesh["first_tri"]), int(mesh["first_tri"]) + int(mesh["tri_count"])):
            ia, ib, ic = (int(k) for k in tris[j]["indices"])
            ab, ac = pos[ib] - pos[ia], pos[ic] - pos[ia]
            norm = cross(ab, ac)
            ao = o - pos[ia]
            dao = cross(ao, d)
            with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
                det = -dot(d, norm[None, :])
                inv_det = f32(1.0) / det
                dst = dot(ao, norm[None, :]) * inv_det
                u = dot(ac[None, :], dao) * inv_det
                v = -dot(ab[None, :], dao) * inv_det
                w = f32(1.0) - u - v
                ok = inside & (det >= f32(1e-6)) & (dst >= 0) & (u >= 0) & (v >= 0) & (w >= 0)
                take = ok & (~hit | (t > dst))                      # strict: the earlier triangle keeps a tie (wgsl:164)
                if take.any():
                    nn = normalize(nrm[ia][None, :] * w[:, None] + nrm[ib][None, :] * u[:, None] + nrm[ic][None, :] * v[:, None])
                    pp = o + d * dst[:, None]
                    hit |= take
                    t = np.where(take, dst, t)
                    prim = np.where(take, np.uint32(j), prim)
                    mat = np.where(take, np.uint32(mesh["material"]), mat)
                    point = np.where(take[:, None], pp, point)
                    normal = np.where(take[:, None], nn, normal)
    return hit, t, prim, mat, point, normal


def shade(hit, mat, normal, materials):
    """wgsl:203-211: debug grey for material >= 100, else colour * dot(normal, +Y) in all four channels; miss = vec4(0)."""
    n = hit.shape[0]
    out = np.zeros((n, 4), f32)
    ndl = dot(normal, np.array([0, 1, 0], f32)[None, :])
    cols = np.zeros((n, 4), f32)
    in_range = mat < len(materials)
    cols[in_range] = materials["color"].astype(f32)[mat[in_range]]
    lit = cols * ndl[:, None]
    grey = f32(0.5) + (mat.astype(np.int64) - 100).astype(f32) / f32(6.0)
    out = np.where((mat >= 100)[:, None], grey[:, None] * np.ones(4, f32), lit)
    return np.where(hit[:, None], out, f32(0.0)).astype(f32)


@pytest.mark.parametrize("name", ["default_64x36_shadow", "spheres8_48x32"])
def test_numpy_restatement_agrees_with_the_c_oracle_and_the_golden_frames(name):
    g = np.load(os.path.join(HERE, "golden", name + ".npz"))
    W, H = (int(x) for x in g["size"])
    o, d = ray_gen(g["world_from_clip"].astype(f32), g["world_position"], W, H)
    tris, meshes, verts, mats = g["triangles"], g["meshes"], g["vertices"], g["materials"]
    hit, t, prim, mat, point, normal = hit_triangles(o, d, tris, meshes, verts)
    rgba = shade(hit, mat, normal, mats)
    shadow = bool(int(g["flags"]) & orc.FLAG_SHADOW)
    if shadow:       # build-defined: origin = point + normal * 1e-3, direction +Y, occluded => rgba * 0.25
        so = point + normal * f32(1e-3)
        sd = np.broadcast_to(np.array([0, 1, 0], f32), so.shape).copy()
        # a ray that missed casts no shadow ray; give it an origin from which nothing can be hit
        occluded, *_ = hit_triangles(np.where(hit[:, None], so, f32(np.nan)), sd, tris, meshes, verts)
        rgba = np.where((hit & occluded)[:, None], rgba * f32(0.25), rgba)
    prim_img = np.where(hit, prim, np.uint32(0xFFFFFFFF)).reshape(H, W)
    t_img = np.where(hit, t, f32(0.0)).reshape(H, W)
    rgba_img = rgba.reshape(H, W, 4)
    # the committed golden frame (written by the C oracle, reproduced bit for bit by every GPU kernel variant)
    assert prim_img.tobytes() == g["prim"].tobytes()
    assert t_img.tobytes() == g["t"].tobytes()
    assert rgba_img.tobytes() == g["rgba"].tobytes()
    # and the C oracle as it is built today
    scene = orc.Scene(tris, meshes, verts, mats)
    view = orc.make_view(g["world_from_clip"], g["world_position"], W, H)
    rgba_o, prim_o, t_o, _ = orc.render_primary(scene, view, orc.FLAG_SHADOW if shadow else 0, threads=2)
    assert prim_img.tobytes() == prim_o.tobytes() and t_img.tobytes() == t_o.tobytes() and rgba_img.tobytes() == rgba_o.tobytes()


# ---- the path-tracing contract (loop shape of wgsl:180-202; build-defined parts: DESIGN.md section 2) -----------------
u32 = np.uint32


def pcg_hash(x):
    with np.errstate(over="ignore"):
        st = x * u32(747796405) + u32(2891336453)
        word = ((st >> ((st >> u32(28)) + u32(4))) ^ st) * u32(277803737)
    return (word >> u32(22)) ^ word


def next_random(state):            # wgsl:39-43; returns (new state, value)
    with np.errstate(over="ignore"):
        state = state * u32(747796405) + u32(2891336453)
        word = ((state >> ((state >> u32(28)) + u32(4))) ^ state) * u32(277803737)
    return state, (word >> u32(22)) ^ word


def rand(state):                   # wgsl:53-55
    state, v = next_random(state)
    return state, v.astype(f32) / f32(4294967295.0)


def det_logf(x):
    x = x.astype(f32)
    out = np.empty_like(x)
    bad = np.isnan(x) | (x < 0)
    zero = x == 0
    ix = x.view(u32).copy()
    k = np.zeros(x.shape, np.int32)
    sub = (ix < u32(0x00800000)) & ~zero & ~bad
    xs = np.where(sub, x * f32(8388608.0), x).astype(f32)
    ix = xs.view(u32).copy()
    k = np.where(sub, -23, 0).astype(np.int32)
    inf = ix >= u32(0x7F800000)
    with np.errstate(over="ignore", invalid="ignore", divide="ignore"):
        ix = ix + u32(0x3F800000 - 0x3F3504F3)
        k = k + (ix >> u32(23)).astype(np.int32) - 127
        ix = (ix & u32(0x007FFFFF)) + u32(0x3F3504F3)
        m = ix.view(f32)
        f = m - f32(1.0)
        s = f / (f32(2.0) + f)
        z = s * s
        w = z * z
        t1 = w * (f32(0.40000972152) + w * f32(0.24279078841))
        t2 = z * (f32(0.66666662693) + w * f32(0.28498786688))
        R = t2 + t1
        hfsq = f32(0.5) * f * f
        dk = k.astype(f32)
        out = dk * f32(6.9313812256e-01) + ((s * (hfsq + R) + dk * f32(9.0580006145e-06)) - hfsq + f)
    out = np.where(inf, xs, out)
    out = np.where(zero, f32(-np.inf), out)
    return np.where(bad, f32(np.nan), out).astype(f32)


def det_cosf(x):
    x = x.astype(f32)
    with np.errstate(invalid="ignore"):
        bad = np.isnan(x) | ((x - x) != 0)
        ax = np.where(x < 0, -x, x)
        q = np.floor(ax * f32(0.63661977237) + f32(0.5)).astype(f32)
        r = ((ax - q * f32(1.5703125)) - q * f32(4.837512969970703125e-4)) - q * f32(7.54978995489188e-8)
        n = np.where(bad, 0, q).astype(np.int64) & 3
        r2 = r * r
        sp = r + r * r2 * (f32(-1.6666654611e-1) + r2 * (f32(8.3321608736e-3) + r2 * f32(-1.9515295891e-4)))
        cp = (f32(1.0) - f32(0.5) * r2) + r2 * r2 * (f32(4.166664568298827e-2) + r2 * (f32(-1.388731625493765e-3) + r2 * f32(2.443315711809948e-5)))
    out = np.where(n == 0, cp, np.where(n == 1, -sp, np.where(n == 2, -cp, sp)))
    return np.where(bad, f32(np.nan), out).astype(f32)


def rand_norm(state):              # wgsl:57-61; theta's rand() is drawn first
    state, r1 = rand(state)
    theta = f32(2.0) * f32(3.1415926) * r1
    state, r2 = rand(state)
    with np.errstate(invalid="ignore", divide="ignore"):
        rho = np.sqrt(f32(-2.0) * det_logf(r2))
        return state, (rho * det_cosf(theta)).astype(f32)


def rand_hemi(state, n):           # wgsl:63-74
    state, x = rand_norm(state)
    state, y = rand_norm(state)
    state, z = rand_norm(state)
    with np.errstate(invalid="ignore", divide="ignore"):
        d = normalize(np.stack([x, y, z], axis=-1))
        flip = dot(n, d) < 0
    return state, np.where(flip[:, None], -d, d).astype(f32)


def test_numpy_restatement_of_the_path_loop_reproduces_the_golden_running_sums():
    g = np.load(os.path.join(HERE, "golden", "default_64x36_shadow.npz"))
    W, H = (int(x) for x in g["size"])
    spp, seed, segs = int(g["path_spp"]), int(g["path_seed"]), 4
    tris, meshes, verts, mats = g["triangles"], g["meshes"], g["vertices"], g["materials"]
    o0, d0 = ray_gen(g["world_from_clip"].astype(f32), g["world_position"], W, H)
    n = W * H
    pixel = np.arange(n, dtype=u32)
    acc = np.zeros((n, 4), f32)
    colours = mats["color"].astype(f32)[:, :3]
    total_segments = 0
    for k in range(spp):
        with np.errstate(over="ignore"):                        # the integer seed: pixel, sample, seed, stream
            h = pcg_hash(pixel)
            h = pcg_hash(h ^ u32((k * 0x9E3779B9 + 0x7F4A7C15) & 0xFFFFFFFF))
            h = pcg_hash(h ^ u32((seed + 0x85EBCA6B) & 0xFFFFFFFF))
            state = pcg_hash(h ^ u32((0 * 0xC2B2AE35 + 0x27D4EB2F) & 0xFFFFFFFF))
        state, jx = rand(state)
        state, jy = rand(state)
        ro = o0.copy()
        rd = (d0 + np.stack([jx, jy, np.zeros(n, f32)], axis=-1) * f32(0.0015)).astype(f32)     # wgsl:184, not renormalised
        col = np.ones((n, 3), f32)
        light = np.zeros((n, 3), f32)
        live = np.ones(n, bool)
        for _ in range(segs):
            idx = np.flatnonzero(live)
            if idx.size == 0:
                break
            total_segments += idx.size
            hit, t, prim, mat, point, normal = hit_triangles(ro[idx], rd[idx], tris, meshes, verts)
            miss = idx[~hit]
            light[miss] = light[miss] + np.array([0.5, 0.71, 0.86], f32)[None, :] * col[miss]          # sky, wgsl:197; break
            live[miss] = False
            hi = idx[hit]
            st, nd = rand_hemi(state[hi], normal[hit])
            state[hi] = st
            ro[hi], rd[hi] = point[hit], nd
            m = mat[hit]
            cm = np.zeros((hi.size, 3), f32)
            ok = m < len(mats)
            cm[ok] = colours[m[ok]]
            col[hi] = np.where(ok[:, None], col[hi] * cm, f32(0.0))
        acc[:, :3] = acc[:, :3] + light
        acc[:, 3] = acc[:, 3] + f32(1.0)
    assert acc.reshape(H, W, 4).tobytes() == g["path_accum"].tobytes()
    assert total_segments == int(g["path_segments"])
