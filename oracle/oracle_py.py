"""ctypes view of oracle/liboracle.so -- the CPU ORACLE (test infrastructure, see rt_oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
The product package (ray-tracing_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")

FLAG_SHADOW = 1
FLAG_NO_MESH_CULL = 2
MISS_PRIM = 0xFFFFFFFF

TRIANGLE = np.dtype([("indices", "<u4", 3)])
MESH_INFO = np.dtype([("first_tri", "<u4"), ("tri_count", "<u4"), ("material", "<u4"), ("_pad0", "<u4"),
                      ("aabb_min", "<f4", 3), ("_pad1", "<f4"), ("aabb_max", "<f4", 3), ("_pad2", "<f4")])
VERTEX = np.dtype([("pos", "<f4", 3), ("_pad0", "<f4"), ("norm", "<f4", 3), ("_pad1", "<f4")])
MATERIAL = np.dtype([("color", "<f4", 4)])
SPHERE = np.dtype([("center", "<f4", 3), ("radius", "<f4")])   # extension: analytic spheres (rt_oracle.h)


class OScene(C.Structure):
    _fields_ = [("tris", C.c_void_p), ("n_tri", C.c_size_t), ("meshes", C.c_void_p), ("n_mesh", C.c_size_t),
                ("verts", C.c_void_p), ("n_vtx", C.c_size_t), ("mats", C.c_void_p), ("n_mat", C.c_size_t),
                ("spheres", C.c_void_p), ("sphere_mats", C.c_void_p), ("n_sphere", C.c_size_t)]


class OView(C.Structure):
    _fields_ = [("world_from_clip", C.c_float * 16), ("world_position", C.c_float * 3),
                ("width", C.c_uint32), ("height", C.c_uint32),
                ("vp_x", C.c_uint32), ("vp_y", C.c_uint32), ("vp_w", C.c_uint32), ("vp_h", C.c_uint32)]


class OHit(C.Structure):
    _fields_ = [("hit", C.c_int), ("point", C.c_float * 3), ("normal", C.c_float * 3), ("t", C.c_float),
                ("material", C.c_uint32), ("prim", C.c_uint32)]


class OTree(C.Structure):
    _fields_ = [("nodes", C.c_void_p), ("n_nodes", C.c_size_t), ("wide", C.c_void_p), ("n_wide", C.c_size_t),
                ("recs", C.c_void_p), ("n_recs", C.c_size_t)]


class OWalkCounts(C.Structure):
    _fields_ = [("node_visits", C.c_uint64), ("tri_tests", C.c_uint64), ("mesh_culls", C.c_uint64)]


class OEntity(C.Structure):
    _fields_ = [("positions", C.c_void_p), ("normals", C.c_void_p), ("n_vtx", C.c_uint32),
                ("indices", C.c_void_p), ("n_idx", C.c_uint32),
                ("base_color_linear", C.c_float * 4), ("scale", C.c_float * 3), ("rotation", C.c_float * 4),
                ("translation", C.c_float * 3), ("aabb_half_extents", C.c_float * 3)]


_lib = None


def build():
    subprocess.run(["make", "-C", _HERE], check=True, stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        build()
    L = C.CDLL(LIB_PATH)
    fp, u32p = C.POINTER(C.c_float), C.POINTER(C.c_uint32)
    L.orc_ray_gen.argtypes = [C.POINTER(OView), C.c_uint32, C.c_uint32, fp, fp]
    L.orc_ray_aabb.argtypes = [fp, fp, fp, fp]
    L.orc_ray_aabb.restype = C.c_int
    L.orc_ray_triangle.argtypes = [C.POINTER(OScene), fp, fp, C.c_uint32, C.POINTER(OHit)]
    L.orc_hit_triangles.argtypes = [C.POINTER(OScene), fp, fp, C.c_uint32, C.POINTER(OHit)]
    L.orc_ray_sphere.argtypes = [C.POINTER(OScene), fp, fp, C.c_uint32, C.POINTER(OHit)]
    L.orc_shade.argtypes = [C.POINTER(OScene), C.POINTER(OHit), fp]
    L.orc_render_primary.argtypes = [C.POINTER(OScene), C.POINTER(OView), C.c_uint32, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    L.orc_render_primary.restype = C.c_uint64
    L.orc_hit_tree.argtypes = [C.POINTER(OScene), C.POINTER(OTree), fp, fp, C.c_uint32, C.c_int, C.POINTER(OHit),
                               C.POINTER(OWalkCounts)]
    L.orc_render_primary_tree.argtypes = [C.POINTER(OScene), C.POINTER(OTree), C.POINTER(OView), C.c_uint32, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.POINTER(OWalkCounts)]
    L.orc_render_primary_tree.restype = C.c_uint64
    L.orc_next_random.argtypes = [u32p]
    L.orc_next_random.restype = C.c_uint32
    L.orc_rand.argtypes = [u32p]
    L.orc_rand.restype = C.c_float
    for n in ("orc_det_logf", "orc_det_cosf"):
        getattr(L, n).argtypes = [C.c_float]
        getattr(L, n).restype = C.c_float
    L.orc_rand_norm.argtypes = [u32p]
    L.orc_rand_norm.restype = C.c_float
    L.orc_rand_dir.argtypes = [u32p, fp]
    L.orc_rand_hemi.argtypes = [u32p, fp, fp]
    L.orc_seed.argtypes = [C.c_uint32] * 4
    L.orc_seed.restype = C.c_uint32
    L.orc_render_path.argtypes = [C.POINTER(OScene), C.POINTER(OView), C.c_uint32, C.c_uint32, C.c_uint32,
                                  C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_int]
    L.orc_render_path.restype = C.c_uint64
    L.orc_render_path_tree.argtypes = [C.POINTER(OScene), C.POINTER(OTree), C.POINTER(OView), C.c_uint32, C.c_uint32,
                                       C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_int,
                                       C.POINTER(OWalkCounts)]
    L.orc_render_path_tree.restype = C.c_uint64
    L.orc_render_path_tree_crop.argtypes = [C.POINTER(OScene), C.POINTER(OTree), C.POINTER(OView), C.c_uint32, C.c_uint32,
                                            C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.c_void_p,
                                            C.c_int, C.POINTER(OWalkCounts)]
    L.orc_render_path_tree_crop.restype = C.c_uint64
    L.orc_resolve.argtypes = [C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p]
    L.orc_encode_bgra8_srgb.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    L.orc_prepare_meshinfo.argtypes = [C.POINTER(OEntity), C.c_size_t, C.c_void_p, C.POINTER(C.c_size_t),
                                       C.c_void_p, C.POINTER(C.c_size_t), C.c_void_p, C.POINTER(C.c_size_t),
                                       C.c_void_p, C.POINTER(C.c_size_t)]
    _lib = L
    return L


def _f3(a):
    return (C.c_float * 3)(*[float(x) for x in a])


class Scene:
    def __init__(self, triangles, meshes, vertices, materials, spheres=None, sphere_materials=None):
        self.triangles = np.ascontiguousarray(triangles, dtype=TRIANGLE)
        self.meshes = np.ascontiguousarray(meshes, dtype=MESH_INFO)
        self.vertices = np.ascontiguousarray(vertices, dtype=VERTEX)
        self.materials = np.ascontiguousarray(materials, dtype=MATERIAL)
        self.spheres = np.ascontiguousarray(spheres if spheres is not None else np.zeros(0, SPHERE), dtype=SPHERE)
        self.sphere_materials = np.ascontiguousarray(
            sphere_materials if sphere_materials is not None else np.zeros(self.spheres.size, np.uint32), dtype=np.uint32)
        assert self.sphere_materials.size == self.spheres.size
        self.c = OScene(self.triangles.ctypes.data, self.triangles.size, self.meshes.ctypes.data, self.meshes.size,
                        self.vertices.ctypes.data, self.vertices.size, self.materials.ctypes.data, self.materials.size,
                        self.spheres.ctypes.data, self.sphere_materials.ctypes.data, self.spheres.size)


class Tree:
    """An acceleration tree as DATA (rt_oracle.h "tree walk"): 64-B binary nodes, 48-B leaf records and, optionally,
    64-B wide nodes (given => the wide tree is the one walked).  Arrays of any dtype with those item sizes."""

    def __init__(self, nodes, recs, wide=None):
        self.nodes = np.ascontiguousarray(nodes)
        self.recs = np.ascontiguousarray(recs)
        self.wide = None if wide is None else np.ascontiguousarray(wide)
        assert self.nodes.dtype.itemsize == 64 and self.recs.dtype.itemsize == 48
        assert self.wide is None or self.wide.dtype.itemsize == 64
        self.c = OTree(self.nodes.ctypes.data, self.nodes.size,
                       self.wide.ctypes.data if self.wide is not None else None, self.wide.size if self.wide is not None else 0,
                       self.recs.ctypes.data, self.recs.size)


def hit_tree(scene, tree, o, d, flags=0, any_hit=False):
    """Returns (hit dict, (node_visits, tri_tests, mesh_culls))."""
    h, c = OHit(), OWalkCounts()
    lib().orc_hit_tree(C.byref(scene.c), C.byref(tree.c), _f3(o), _f3(d), flags, int(any_hit), C.byref(h), C.byref(c))
    return _hit(h), (c.node_visits, c.tri_tests, c.mesh_culls)


def render_primary_tree(scene, tree, view, flags=0, threads=8, pixel_list=None):
    """render_primary through the tree walk.  Returns (rgba, prim, t, rays, (node_visits, tri_tests, mesh_culls))."""
    c = OWalkCounts()
    if pixel_list is None:
        H, W = view.height, view.width
        rgba, prim, t = np.zeros((H, W, 4), np.float32), np.zeros((H, W), np.uint32), np.zeros((H, W), np.float32)
        pl_ptr, n = None, 0
    else:
        pl = np.ascontiguousarray(pixel_list, dtype=np.uint32)
        n = pl.size
        rgba, prim, t = np.zeros((n, 4), np.float32), np.zeros(n, np.uint32), np.zeros(n, np.float32)
        pl_ptr = pl.ctypes.data
    rays = lib().orc_render_primary_tree(C.byref(scene.c), C.byref(tree.c), C.byref(view), flags, rgba.ctypes.data,
                                         prim.ctypes.data, t.ctypes.data, pl_ptr, n, threads, C.byref(c))
    return rgba, prim, t, int(rays), (int(c.node_visits), int(c.tri_tests), int(c.mesh_culls))


def make_view(world_from_clip, world_position, width, height, viewport=None):
    v = OView()
    m = np.asarray(world_from_clip, np.float32).ravel()
    p = np.asarray(world_position, np.float32).ravel()
    for k in range(16):
        v.world_from_clip[k] = m[k]
    for k in range(3):
        v.world_position[k] = p[k]
    v.width, v.height = width, height
    if viewport is not None:
        v.vp_x, v.vp_y, v.vp_w, v.vp_h = viewport
    return v


def ray_gen(view, px, py):
    o, d = (C.c_float * 3)(), (C.c_float * 3)()
    lib().orc_ray_gen(C.byref(view), px, py, o, d)
    return np.array(o, np.float32), np.array(d, np.float32)


def ray_aabb(o, d, lb, rt):
    return bool(lib().orc_ray_aabb(_f3(o), _f3(d), _f3(lb), _f3(rt)))


def _hit(h):
    return dict(hit=bool(h.hit), point=np.array(h.point, np.float32), normal=np.array(h.normal, np.float32),
                t=float(np.float32(h.t)), material=h.material, prim=h.prim)


def ray_triangle(scene, o, d, tri):
    h = OHit()
    lib().orc_ray_triangle(C.byref(scene.c), _f3(o), _f3(d), tri, C.byref(h))
    return _hit(h)


def ray_sphere(scene, o, d, sphere):
    h = OHit()
    lib().orc_ray_sphere(C.byref(scene.c), _f3(o), _f3(d), sphere, C.byref(h))
    return _hit(h)


def hit_triangles(scene, o, d, flags=0):
    h = OHit()
    lib().orc_hit_triangles(C.byref(scene.c), _f3(o), _f3(d), flags, C.byref(h))
    return _hit(h)


def shade(scene, hit):
    h = OHit()
    h.hit = int(hit["hit"])
    for k in range(3):
        h.point[k] = hit["point"][k]
        h.normal[k] = hit["normal"][k]
    h.t, h.material, h.prim = hit["t"], hit["material"], hit["prim"]
    out = (C.c_float * 4)()
    lib().orc_shade(C.byref(scene.c), C.byref(h), out)
    return np.array(out, np.float32)


def render_primary(scene, view, flags=0, threads=8, pixel_list=None):
    """Returns (rgba, prim, t, rays).  Full frame: arrays shaped (H, W, ...); with pixel_list: (n, ...)."""
    if pixel_list is None:
        H, W = view.height, view.width
        rgba, prim, t = np.zeros((H, W, 4), np.float32), np.zeros((H, W), np.uint32), np.zeros((H, W), np.float32)
        rays = lib().orc_render_primary(C.byref(scene.c), C.byref(view), flags, rgba.ctypes.data, prim.ctypes.data,
                                        t.ctypes.data, None, 0, threads)
    else:
        pl = np.ascontiguousarray(pixel_list, dtype=np.uint32)
        n = pl.size
        rgba, prim, t = np.zeros((n, 4), np.float32), np.zeros(n, np.uint32), np.zeros(n, np.float32)
        rays = lib().orc_render_primary(C.byref(scene.c), C.byref(view), flags, rgba.ctypes.data, prim.ctypes.data,
                                        t.ctypes.data, pl.ctypes.data, n, threads)
    return rgba, prim, t, int(rays)


def render_path(scene, view, spp, max_segments=4, seed=0, stream=0, sample_first=0, flags=0, accum=None, threads=8):
    H, W = view.height, view.width
    if accum is None:
        accum = np.zeros((H, W, 4), np.float32)
    segs = lib().orc_render_path(C.byref(scene.c), C.byref(view), flags, spp, sample_first, max_segments, seed,
                                 stream, accum.ctypes.data, threads)
    return accum, int(segs)


def render_path_tree(scene, tree, view, spp, max_segments=4, seed=0, stream=0, sample_first=0, flags=0, accum=None, threads=8):
    """render_path through the tree walk.  Returns (accum, segments, (node_visits, tri_tests, mesh_culls))."""
    H, W = view.height, view.width
    if accum is None:
        accum = np.zeros((H, W, 4), np.float32)
    c = OWalkCounts()
    segs = lib().orc_render_path_tree(C.byref(scene.c), C.byref(tree.c), C.byref(view), flags, spp, sample_first,
                                      max_segments, seed, stream, accum.ctypes.data, threads, C.byref(c))
    return accum, int(segs), (int(c.node_visits), int(c.tri_tests), int(c.mesh_culls))


def render_path_tree_crop(scene, tree, view, crop, spp, max_segments=4, seed=0, stream=0, sample_first=0, flags=0, threads=8):
    """The running sums of render_path_tree for the pixels of crop = (x, y, w, h) only, returned as an (h, w, 4) array:
    a pixel's value does not depend on the crop (whole frames at 4K x 64 samples or 8K take the host minutes)."""
    x, y, w, h = crop
    H, W = view.height, view.width
    assert x + w <= W and y + h <= H
    # the C side indexes the whole target: give it a buffer whose rows y .. y + h - 1 exist and point it h rows up
    rows = np.zeros((h, W, 4), np.float32)
    base = rows.ctypes.data - y * W * 16
    c = OWalkCounts()
    lib().orc_render_path_tree_crop(C.byref(scene.c), C.byref(tree.c), C.byref(view), flags, spp, sample_first, max_segments,
                                    seed, stream, (C.c_uint32 * 4)(x, y, w, h), base, threads, C.byref(c))
    return rows[:, x:x + w].copy()


def resolve(accum, total_samples):
    out = np.empty_like(accum)
    lib().orc_resolve(accum.ctypes.data, accum.size // 4, total_samples, out.ctypes.data)
    return out


def encode_bgra8_srgb(rgba):
    rgba = np.ascontiguousarray(rgba, np.float32)
    out = np.empty(rgba.shape[:-1] + (4,), np.uint8)
    lib().orc_encode_bgra8_srgb(rgba.ctypes.data, rgba.size // 4, out.ctypes.data)
    return out


def prepare_meshinfo(entities):
    """entities: list of dicts as ray_tracing_amd.App.entities() returns."""
    n = len(entities)
    arr = (OEntity * max(n, 1))()
    keep = []
    tot_v = tot_i = 0
    for k, e in enumerate(entities):
        E = arr[k]
        p = None if e["positions"] is None else np.ascontiguousarray(e["positions"], np.float32)
        nn = None if e["normals"] is None else np.ascontiguousarray(e["normals"], np.float32)
        ix = None if e["indices"] is None else np.ascontiguousarray(e["indices"], np.uint32)
        keep += [p, nn, ix]
        E.positions = p.ctypes.data if p is not None else None
        E.normals = nn.ctypes.data if nn is not None else None
        E.n_vtx = (p.size // 3) if p is not None else 0
        E.indices = ix.ctypes.data if ix is not None else None
        E.n_idx = ix.size if ix is not None else 0
        for j in range(4):
            E.base_color_linear[j] = e["base_color_linear"][j]
            E.rotation[j] = e["rotation"][j]
        for j in range(3):
            E.scale[j] = e["scale"][j]
            E.translation[j] = e["translation"][j]
            E.aabb_half_extents[j] = e["aabb_half_extents"][j]
        tot_v += E.n_vtx
        tot_i += E.n_idx
    t, m = np.zeros(tot_i // 3 + 1, TRIANGLE), np.zeros(n + 1, MESH_INFO)
    v, c = np.zeros(tot_v + 1, VERTEX), np.zeros(n + 1, MATERIAL)
    nt, nm, nv, nc = C.c_size_t(), C.c_size_t(), C.c_size_t(), C.c_size_t()
    lib().orc_prepare_meshinfo(arr, n, t.ctypes.data, C.byref(nt), m.ctypes.data, C.byref(nm), v.ctypes.data,
                               C.byref(nv), c.ctypes.data, C.byref(nc))
    return Scene(t[:nt.value], m[:nm.value], v[:nv.value], c[:nc.value])
