"""Experiment script (GPU): rt_update_vertices on the bench scene, GPU refit vs host refit + re-upload. Not a test."""
import os, sys, time
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [_ROOT]
import numpy as np, ray_tracing_amd as rt
app = rt.App(); app.spawn_sphere_field(10000, 0x5EED0002, 5); app.camera_frame_scene(1.0); app.camera_set_target(1920, 1080)
sc = app.prepare_meshinfo(); m, p = app.camera_view()
moved = rt.Scene(sc.triangles, sc.meshes.copy(), sc.vertices.copy(), sc.materials)
moved.vertices["pos"][:, 1] += 0.25; moved.meshes["aabb_min"][:, 1] += 0.25; moved.meshes["aabb_max"][:, 1] += 0.25
# the dirty-mesh form: one sphere (mesh 4321) of the 10,000 moves
one = rt.Scene(sc.triangles, sc.meshes.copy(), sc.vertices.copy(), sc.materials)
k = 4321
mi = one.meshes[k]
vs = np.unique(one.triangles["indices"][mi["first_tri"]:mi["first_tri"] + mi["tri_count"]])
one.vertices["pos"][vs, 1] += 0.25; one.meshes["aabb_min"][k, 1] += 0.25; one.meshes["aabb_max"][k, 1] += 0.25
ctx = rt.Context(0); ctx.upload_scene(sc); ctx.set_view(m, p, 1920, 1080)
for rep in range(3):
    t0 = time.perf_counter(); ctx.update_vertex_range(one, int(vs.min()), int(vs.max() - vs.min() + 1), k, 1); dt = time.perf_counter() - t0
    print("range rt_update_vertex_range (1 of 10,000 spheres, %d vertices) wall %.2f ms, device %.2f ms" % (len(vs), dt * 1e3, ctx.bvh_info().build_ms), flush=True)
ctx.render(flags=rt.FLAG_SHADOW); prim_r = ctx.read_prim_ids()
# more meshes at once, the three device forms: ONE launch of one workgroup over a host-made node list / one launch per tree
# level over the listed nodes of that level / per-level launches over the whole tree with a dirty byte per node
for name, lim, lev in (("one workgroup", "16777216", "0"), ("level lists", "0", "67108864"), ("whole tree", "0", "0"), ("defaults", None, None)):
    for key, val in (("RT_TUNE_REFIT_LIST_MAX", lim), ("RT_TUNE_REFIT_LEVELS_MAX", lev)):
        if val is None: os.environ.pop(key, None)
        else: os.environ[key] = val
    cx = rt.Context(0); cx.upload_scene(sc); cx.set_view(m, p, 1920, 1080)
    for nm in (1, 10, 50, 200, 1000):
        k0 = 4000
        vlo = int(min(np.min(sc.triangles["indices"][sc.meshes[q]["first_tri"]:sc.meshes[q]["first_tri"] + sc.meshes[q]["tri_count"]]) for q in range(k0, k0 + nm)))
        vhi = int(max(np.max(sc.triangles["indices"][sc.meshes[q]["first_tri"]:sc.meshes[q]["first_tri"] + sc.meshes[q]["tri_count"]]) for q in range(k0, k0 + nm)))
        calls = []
        for rep in range(5):     # the same range five times in a row: first call, first repeat (a long list is made now), kept list
            t0 = time.perf_counter(); cx.update_vertex_range(sc, vlo, vhi - vlo + 1, k0, nm); dt = time.perf_counter() - t0
            calls.append((dt * 1e3, cx.bvh_info().build_ms))
        print("%-13s: %4d meshes (%d vertices): wall ms by call %s, device %.2f ms" % (name, nm, vhi - vlo + 1, " ".join("%.2f" % w for w, _ in calls), calls[-1][1]), flush=True)
    cx.close()
ctx2 = rt.Context(0); ctx2.upload_scene(sc); ctx2.set_view(m, p, 1920, 1080); ctx2.update_vertices(one); ctx2.render(flags=rt.FLAG_SHADOW)
print("range == full update:", bool((ctx2.read_prim_ids() == prim_r).all()), flush=True)
ctx.close(); ctx2.close()
for mode in ("gpu", "host"):
    if mode == "host": os.environ["RT_TUNE_HOST_REFIT"] = "1"
    ctx = rt.Context(0); ctx.upload_scene(sc); ctx.set_view(m, p, 1920, 1080)
    t0 = time.perf_counter(); ctx.update_vertices(moved); dt = time.perf_counter() - t0
    ctx.render(flags=rt.FLAG_SHADOW); prim = ctx.read_prim_ids()
    print(mode, "rt_update_vertices wall %.1f ms, refit itself %.2f ms" % (dt * 1e3, ctx.bvh_info().build_ms), "hash", int(prim.astype(np.uint64).sum()), flush=True)
    ctx.close()
