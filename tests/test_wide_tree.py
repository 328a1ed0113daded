"""The 4-wide quantised tree (ray-tracing_amd/csrc/rt_wide.h, bvh_build.cpp rt_bvh_widen) on the CPU: its boxes
contain the binary tree's, every leaf record is reached exactly once, the stack bound holds, and a walk that
uses the kernel's arithmetic (byte*A - B with the clamped 1/d) reaches every leaf the binary walk reaches --
the property that makes the results independent of which tree is walked.  No reference counterpart
(src/bvh.rs:1 is empty): build-side design, parity unpinned by the reference."""
import sys

import numpy as np
import pytest

import ray_tracing_amd as rt
from helpers import default_app, soup_app

NONE = -2**31
EMPTY = 0xFFFFFFFF
f32 = np.float32


def _byte(word, k):
    return (int(word) >> (8 * k)) & 255


def _check_wide(nodes, recs, wn, ws, need):
    cov = np.zeros(len(recs), int)
    seen = np.zeros(len(wn), bool)
    deepest = 0
    for w in range(len(wn)):
        W = wn[w]
        for k in range(4):
            s = int(ws[w]["slot"][k])
            c = int(W["child"][k])
            if s == EMPTY:
                assert c == NONE
                assert all(_byte(W["lo"][a], k) == 255 and _byte(W["hi"][a], k) == 0 for a in range(3))   # inverted box
                continue
            N = nodes[s >> 1]
            lo, hi = (N["lo1"], N["hi1"]) if s & 1 else (N["lo0"], N["hi0"])
            assert c == int(N["child1"] if s & 1 else N["child0"]) or c >= 0   # leaves keep their code, inner children are re-indexed
            for a in range(3):
                if not np.isfinite(W["scl"][a]):
                    assert _byte(W["lo"][a], k) == 0 and _byte(W["hi"][a], k) == 255      # unbounded axis
                    continue
                pl = np.float64(W["org"][a]) + _byte(W["lo"][a], k) * np.float64(W["scl"][a])
                ph = np.float64(W["org"][a]) + _byte(W["hi"][a], k) * np.float64(W["scl"][a])
                assert pl <= lo[a] and ph >= hi[a], (w, k, a)
                # ... and not by much: at most one grid step (+ the 2e-6 expansion) beyond the source box
                step = np.float64(W["scl"][a]) * 1.01 + 4e-6 * max(abs(lo[a]), abs(hi[a])) + 4e-12
                assert pl >= lo[a] - step and ph <= hi[a] + step, (w, k, a)
            if c >= 0:
                assert c > w and not seen[c]        # children are stored after their parent, one parent each
                seen[c] = True
            else:
                first, cnt = (~c) >> 3, ((~c) & 7) + 1
                cov[first:first + cnt] += 1
    assert (cov == 1).all() and seen[1:].all()

    # the stack bound: walk the tree the worst way (neediest subtree first) and compare
    def need_of(w):
        sub = sorted((need_of(int(c)) if c >= 0 else 0 for c in wn[w]["child"] if c != NONE), reverse=True)
        return max((s + len(sub) - 1 - i for i, s in enumerate(sub)), default=0)
    if len(wn):
        assert need_of(0) == need
    return deepest


def _walk_binary(nodes, o, d):
    with np.errstate(all="ignore"):
        inv = f32(1) / d
    stack, leaves = [0], set()
    while stack:
        c = stack.pop()
        if c < 0:
            leaves.add(c)
            continue
        N = nodes[c]
        for lo, hi, ch in ((N["lo0"], N["hi0"], int(N["child0"])), (N["lo1"], N["hi1"], int(N["child1"]))):
            if ch == NONE:
                continue
            with np.errstate(all="ignore"):
                t0, t1 = (lo - o) * inv, (hi - o) * inv
            tmin = np.fmax(np.fmax(np.fmin(t0[0], t1[0]), np.fmin(t0[1], t1[1])), np.fmin(t0[2], t1[2]))
            tmax = np.fmin(np.fmin(np.fmax(t0[0], t1[0]), np.fmax(t0[1], t1[1])), np.fmax(t0[2], t1[2])) * f32(1.0000005)
            if tmin <= tmax and tmax >= 0:
                stack.append(ch)
    return leaves


def _walk_wide(wn, o, d):
    """wide_decide of rt_kernels.hip: A = scl*inv, B = (o-org)*inv, t = fma(byte, A, -B), 1/d clamped to +-1e30."""
    with np.errstate(all="ignore"):
        inv = np.clip(f32(1) / d, f32(-1e30), f32(1e30)).astype(f32)
    stack, leaves, visits, deepest = [0], set(), 0, 0
    while stack:
        deepest = max(deepest, len(stack))
        c = stack.pop()
        if c < 0:
            leaves.add(c)
            continue
        visits += 1
        W = wn[c]
        with np.errstate(all="ignore"):
            A = (W["scl"] * inv).astype(f32)
            B = ((o - W["org"]) * inv).astype(f32)
        for k in range(4):
            ch = int(W["child"][k])
            tn, tf = f32(0), f32(np.inf)
            for a in range(3):
                bl, bh = _byte(W["lo"][a], k), _byte(W["hi"][a], k)
                n_, f_ = (bh, bl) if inv[a] < 0 else (bl, bh)
                with np.errstate(all="ignore"):        # fma: one rounding of the exact product-sum
                    tn = np.fmax(tn, f32(np.float64(n_) * np.float64(A[a]) - np.float64(B[a])))
                    tf = np.fmin(tf, f32(np.float64(f_) * np.float64(A[a]) - np.float64(B[a])))
            if tn <= tf * f32(1.000001) and ch != NONE:
                stack.append(ch)
    return leaves, visits, deepest


def _scenes():
    yield "default", default_app().prepare_meshinfo()
    yield "soup", soup_app(16, 0x5EED0003, 64, 64, subdivisions=2).prepare_meshinfo()


@pytest.mark.parametrize("name", ["default", "soup"])
def test_wide_tree_contains_the_binary_boxes_and_covers_every_record(name):
    sys.setrecursionlimit(10000)
    scene = dict(_scenes())[name]
    info, nodes, recs, wn, ws, need = rt.host_bvh_build(scene, threads=2, wide=True)
    assert 0 < len(wn) <= 0.6 * len(nodes) + 1
    n_children = int((ws["slot"] != EMPTY).sum())
    assert n_children / len(wn) > 2.8                       # the greedy collapse fills most slots
    _check_wide(nodes, recs, wn, ws, need)
    assert need + 1 <= 80                                    # RT_STACK_CAP of rt_layout.h


def test_wide_tree_after_a_refit_contains_the_moved_boxes():
    sys.setrecursionlimit(10000)
    app = default_app()
    scene = app.prepare_meshinfo()
    app.rotate_tagged(0.7)
    moved = app.prepare_meshinfo()
    info, nodes, recs, wn, ws, need = rt.host_bvh_build(scene, threads=2, refit_to=moved, wide=True)
    _check_wide(nodes, recs, wn, ws, need)


def test_wide_walk_reaches_every_leaf_the_binary_walk_reaches():
    scene = soup_app(16, 0x5EED0003, 64, 64, subdivisions=2).prepare_meshinfo()
    info, nodes, recs, wn, ws, need = rt.host_bvh_build(scene, threads=2, wide=True)
    rng = np.random.default_rng(5)
    lo = np.minimum(nodes[0]["lo0"], nodes[0]["lo1"])
    hi = np.maximum(nodes[0]["hi0"], nodes[0]["hi1"])
    total_b = total_w = 0
    for i in range(160):
        o = rng.uniform(lo - 5, hi + 5).astype(f32)
        if i % 4 == 0:        # axis-parallel rays: zero components, infinite reciprocals (the shadow rays are (0,1,0))
            d = np.zeros(3, f32)
            d[rng.integers(3)] = f32(rng.choice([-1.0, 1.0]))
            if i % 8 == 0:
                d[(int(np.flatnonzero(d)[0]) + 1) % 3] = f32(1e-39)      # a subnormal component: 1/d overflows
        elif i % 4 == 1:      # origins far outside the scene
            o = (o * f32(1000.0)).astype(f32)
            d = (rng.uniform(lo, hi).astype(f32) - o)
            d = (d / np.linalg.norm(d)).astype(f32)
        else:
            d = rng.uniform(lo, hi).astype(f32) - o
            d = (d / np.linalg.norm(d)).astype(f32)
        a = _walk_binary(nodes, o, d)
        b, visits, deepest = _walk_wide(wn, o, d)
        assert a <= b, i
        assert deepest <= need + 1
        total_b += len(a)
        total_w += len(b)
    assert total_w <= 1.5 * total_b + 50          # ... without giving the culling away


def test_wide_tree_edge_cases():
    empty = rt.Scene(np.zeros(0, rt.TRIANGLE), np.zeros(0, rt.MESH_INFO), np.zeros(0, rt.VERTEX), np.zeros(0, rt.MATERIAL))
    info, nodes, recs, wn, ws, need = rt.host_bvh_build(empty, wide=True)
    assert len(wn) == 0 and need == 0
    v = np.zeros(3, rt.VERTEX)
    v["pos"] = [(0, 0, 0), (0, 0, 1), (1, 0, 0)]             # flat in y, at the origin
    t = np.zeros(1, rt.TRIANGLE)
    t["indices"] = (0, 1, 2)
    m = np.zeros(1, rt.MESH_INFO)
    m["tri_count"] = 1
    one = rt.Scene(t, m, v, np.zeros(1, rt.MATERIAL))
    info, nodes, recs, wn, ws, need = rt.host_bvh_build(one, wide=True)
    assert len(wn) == 1 and need == 0 and list(wn[0]["child"][1:]) == [NONE] * 3 and wn[0]["child"][0] < 0
    _check_wide(nodes, recs, wn, ws, need)
    # coordinates whose extent overflows f32: the axis falls back to "unbounded" instead of a wrong box
    v2 = np.zeros(6, rt.VERTEX)
    v2["pos"] = [(-3e38, 0, 0), (-3e38, 1, 0), (-3e38, 0, 1), (3e38, 0, 0), (3e38, 1, 0), (3e38, 0, 1)]
    t2 = np.zeros(2, rt.TRIANGLE)
    t2["indices"] = [(0, 1, 2), (3, 4, 5)]
    m2 = np.zeros(1, rt.MESH_INFO)
    m2["tri_count"] = 2
    far = rt.Scene(t2, m2, v2, np.zeros(1, rt.MATERIAL))
    info, nodes, recs, wn, ws, need = rt.host_bvh_build(far, wide=True)
    assert len(wn) == 1 and not np.isfinite(wn[0]["scl"][0]) and np.isfinite(wn[0]["scl"][1:]).all()
    _check_wide(nodes, recs, wn, ws, need)
