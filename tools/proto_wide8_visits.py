import sys, time, numpy as np
sys.path[:0] = ["/root/repo", "/root/repo/oracle"]
import ray_tracing_amd as rt, oracle_py as orc
N_SPH = int(sys.argv[1]) if len(sys.argv) > 1 else 1389
app = rt.App(); app.spawn_icosphere_soup(N_SPH, 0x5EED0003, 5); app.camera_frame_scene(1.0); app.camera_set_target(1920, 1080)
sc = app.prepare_meshinfo(); m, p = app.camera_view()
info, nodes, recs = rt.host_bvh_build(sc, threads=8, wide=False)
print("binary nodes", len(nodes), "recs", len(recs))
lo0, hi0, lo1, hi1 = nodes["lo0"].astype(np.float64), nodes["hi0"].astype(np.float64), nodes["lo1"].astype(np.float64), nodes["hi1"].astype(np.float64)
c0, c1 = nodes["child0"], nodes["child1"]
NONE = -2**31
def area(lo, hi):
    e = hi - lo; return e[0]*e[1] + e[1]*e[2] + e[2]*e[0]
def slots_of(n):
    s = []
    if c0[n] != NONE: s.append((int(c0[n]), lo0[n], hi0[n]))
    if c1[n] != NONE: s.append((int(c1[n]), lo1[n], hi1[n]))
    return s
def collapse(K):
    W = []   # list of (los (k,3), his (k,3), codes list)
    W.append(None); stack = [(0, 0)]
    while stack:
        b, w = stack.pop()
        s = slots_of(b)
        while len(s) < K:
            best = -1
            for k, (c, lo, hi) in enumerate(s):
                if c >= 0 and (best < 0 or area(lo, hi) > area(s[best][1], s[best][2])): best = k
            if best < 0: break
            two = slots_of(s[best][0])
            if len(s) - 1 + len(two) > K: break
            s[best:best+1] = two
        codes = []
        for (c, lo, hi) in s:
            if c >= 0:
                idx = len(W); W.append(None); stack.append((c, idx)); codes.append(idx)
            else: codes.append(c)
        W[w] = (np.array([x[1] for x in s]), np.array([x[2] for x in s]), codes)
    return W
ra, rab, rac = recs["a"].astype(np.float64), recs["ab"].astype(np.float64), recs["ac"].astype(np.float64)
def tri(o, d, k):
    ab, ac = rab[k], rac[k]
    n = np.cross(ab, ac); ao = o - ra[k]; dao = np.cross(ao, d)
    det = -np.dot(d, n)
    if det < 1e-6: return None
    inv = 1.0 / det
    t = np.dot(ao, n) * inv; u = np.dot(ac, dao) * inv; v = -np.dot(ab, dao) * inv
    if t >= 0 and u >= 0 and v >= 0 and 1 - u - v >= 0: return t
    return None
def walk(W, o, d, order, any_hit=False):
    inv = 1.0 / np.where(d == 0, 1e-30, d)
    t = np.inf; visits = 0; tris = 0; pushes = 0; hitsum = 0
    stack = []; cur = 0
    while True:
        if cur is None:
            if not stack: break
            cur = stack.pop(); continue
        if cur >= 0:
            visits += 1
            los, his, codes = W[cur]
            t0 = (los - o) * inv; t1 = (his - o) * inv
            tn = np.maximum(np.minimum(t0, t1).max(1), 0.0); tf = np.maximum(t0, t1).min(1)
            hit = (tn <= tf) & (tn <= t)
            idx = np.nonzero(hit)[0]
            hitsum += len(idx)
            if len(idx) == 0: cur = None; continue
            if order == "sorted": idx = idx[np.argsort(tn[idx], kind="stable")]
            else:
                j = idx[np.argmin(tn[idx])]; idx = np.array([j] + [x for x in idx if x != j])
            for x in idx[:0:-1]: stack.append(codes[x]); pushes += 1
            cur = codes[idx[0]]
        else:
            code = ~cur; first, cnt = code >> 3, (code & 7) + 1
            for k in range(first, first + cnt):
                tris += 1
                tt = tri(o, d, k)
                if tt is not None and tt < t:
                    t = tt
                    if any_hit: return visits, tris, pushes, hitsum, t
            cur = None
    return visits, tris, pushes, hitsum, t
view = orc.make_view(m, p, 1920, 1080)
rng = np.random.default_rng(1)
px = rng.integers(0, 1920, 1500); py = rng.integers(0, 1080, 1500)
rays = []
for x, y in zip(px, py):
    o, d = orc.ray_gen(view, int(x), int(y)); rays.append((np.array(o, np.float64), np.array(d, np.float64)))
for K in (4, 8):
    t0 = time.time(); W = collapse(K); print("K", K, "wide nodes", len(W), "children/node %.2f" % (sum(len(w[2]) for w in W) / len(W)), "%.1fs" % (time.time() - t0))
    for order in ("sorted", "nearest"):
        tot = np.zeros(4); sh = np.zeros(4); nsh = 0
        for o, d in rays:
            v, tr, pu, hs, t = walk(W, o, d, order)
            tot += (v, tr, pu, hs)
            if np.isfinite(t):
                so = o + d * t + np.array([0, 1e-3, 0]); r = walk(W, so, np.array([0.0, 1.0, 0.0]), order, any_hit=True); sh += r[:4]; nsh += 1
        n = len(rays)
        print("  %-8s primary: visits/ray %.2f tris %.2f pushes %.2f hit-children/visit %.2f | shadow (%d): visits %.2f tris %.2f" % (
            order, tot[0] / n, tot[1] / n, tot[2] / n, tot[3] / tot[0], nsh, sh[0] / max(nsh, 1), sh[1] / max(nsh, 1)), flush=True)
