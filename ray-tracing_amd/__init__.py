"""ray_tracing_amd -- thin ctypes view of librt_amd.so (include/rt_amd.h, include/rt_host.h).

The product is the shared library: hand-written HIP kernels for gfx950 behind a C ABI, plus a
C++ mirror of the reference's plugin interface (RayTracingPlugin, prepare_meshinfo,
RayTracingPassNode; camms205/ray-tracing src/ray_tracing.rs, src/node.rs, src/pipeline.rs).
This module only marshals numpy arrays across that ABI for tests and bench.py.  It never
computes a result itself and has no CPU fallback: if the library is missing it raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RT_AMD_LIB", os.path.join(_HERE, "librt_amd.so"))   # override: A/B experiments only

# status codes (include/rt_amd.h)
RT_OK, RT_NOT_READY = 0, 1
RT_ERR_ARG, RT_ERR_HIP, RT_ERR_NO_GPU, RT_ERR_SCENE, RT_ERR_OOM = -1, -2, -3, -4, -5

FLAG_SHADOW = 0x0001
FLAG_NO_MESH_CULL = 0x0002
FLAG_BRUTE_FORCE = 0x0004
FLAG_COUNTERS = 0x0008
FLAG_PATH_TRACE = 0x0010
FLAG_ACCUMULATE = 0x0020
FLAG_TIME_KERNELS = 0x0040
FLAG_WAVE_TRAVERSAL = 0x0080
FLAG_STATIC_GRID = 0x0100
FLAG_BINARY_TREE = 0x0200

MISS_PRIM = 0xFFFFFFFF

# T1-T4 layouts, byte-identical to the RayTracingInfo bindings (src/ray_tracing.rs:51-96)
TRIANGLE = np.dtype([("indices", "<u4", 3)])
MESH_INFO = np.dtype([("first_tri", "<u4"), ("tri_count", "<u4"), ("material", "<u4"), ("_pad0", "<u4"),
                      ("aabb_min", "<f4", 3), ("_pad1", "<f4"), ("aabb_max", "<f4", 3), ("_pad2", "<f4")])
VERTEX = np.dtype([("pos", "<f4", 3), ("_pad0", "<f4"), ("norm", "<f4", 3), ("_pad1", "<f4")])
MATERIAL = np.dtype([("color", "<f4", 4)])
SPHERE = np.dtype([("center", "<f4", 3), ("radius", "<f4")])   # extension: analytic spheres (rt_sphere)
assert TRIANGLE.itemsize == 12 and MESH_INFO.itemsize == 48 and VERTEX.itemsize == 32 and MATERIAL.itemsize == 16

BVH_NODE = np.dtype([("lo0", "<f4", 3), ("hi0", "<f4", 3), ("lo1", "<f4", 3), ("hi1", "<f4", 3),
                     ("child0", "<i4"), ("child1", "<i4"), ("pad", "<u4", 2)])
TRI_RECORD = np.dtype([("a", "<f4", 3), ("ab", "<f4", 3), ("ac", "<f4", 3), ("tri", "<u4"), ("mesh", "<u4"), ("seq", "<u4")])
WIDE_NODE = np.dtype([("org", "<f4", 3), ("scl", "<f4", 3), ("lo", "<u4", 3), ("hi", "<u4", 3), ("child", "<i4", 4)])
WIDE_SRC = np.dtype([("slot", "<u4", 4)])
assert BVH_NODE.itemsize == 64 and TRI_RECORD.itemsize == 48 and WIDE_NODE.itemsize == 64 and WIDE_SRC.itemsize == 16


class RenderParams(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in (
        "struct_size", "flags", "frame_count", "spp", "sample_first", "max_segments", "seed", "stream_id",
        "shard_rank", "shard_count", "tile_size", "_reserved")]


class Counters(C.Structure):
    _fields_ = [("rays_primary", C.c_uint64), ("rays_shadow", C.c_uint64), ("rays_bounce", C.c_uint64),
                ("node_visits", C.c_uint64), ("tri_tests", C.c_uint64), ("mesh_aabb_tests", C.c_uint64),
                ("kernel_ms", C.c_double), ("kernel_launches", C.c_uint32), ("_pad", C.c_uint32),
                ("wave_node_iters", C.c_uint64), ("wave_tri_iters", C.c_uint64), ("wave_alive_lanes", C.c_uint64)]


class BvhInfo(C.Structure):
    _fields_ = [("n_nodes", C.c_uint64), ("n_triangles", C.c_uint64), ("n_meshes", C.c_uint64),
                ("n_vertices", C.c_uint64), ("node_bytes", C.c_uint32), ("tri_bytes", C.c_uint32),
                ("max_depth", C.c_uint32), ("max_leaf", C.c_uint32), ("build_ms", C.c_double),
                ("sah_cost", C.c_double), ("n_builds", C.c_uint32), ("n_refits", C.c_uint32),
                ("n_wide_nodes", C.c_uint64), ("wide_max_depth", C.c_uint32), ("wide_stack_need", C.c_uint32)]


class EntityView(C.Structure):
    _fields_ = [("positions", C.POINTER(C.c_float)), ("normals", C.POINTER(C.c_float)), ("n_vertices", C.c_uint32),
                ("indices", C.POINTER(C.c_uint32)), ("n_indices", C.c_uint32),
                ("base_color_linear", C.c_float * 4), ("scale", C.c_float * 3), ("rotation", C.c_float * 4),
                ("translation", C.c_float * 3), ("aabb_center", C.c_float * 3), ("aabb_half_extents", C.c_float * 3)]


# every symbol include/rt_amd.h and include/rt_host.h declare
RT_SYMBOLS = ["rt_create", "rt_destroy", "rt_set_stream", "rt_set_frames_in_flight", "rt_get_stream", "rt_upload_scene", "rt_upload_scene_ex", "rt_update_vertices", "rt_update_vertex_range", "rt_host_bvh_refit", "rt_set_view", "rt_render",
              "rt_synchronize", "rt_read_radiance", "rt_read_prim_ids", "rt_read_hit_t", "rt_read_bgra8_srgb",
              "rt_read_accumulation", "rt_device_buffers", "rt_resolve_accumulation", "rt_get_counters",
              "rt_encode_bgra8_srgb", "rt_present_async", "rt_present_wait", "rt_tile_slot_bytes", "rt_pack_tiles", "rt_unpack_tiles", "rt_render_static",
              "rt_get_bvh_info", "rt_read_bvh", "rt_read_bvh_wide", "rt_host_bvh_build", "rt_host_bvh_copy", "rt_host_bvh_wide", "rt_host_bvh_free", "rt_measure_gather_rates", "rt_last_error", "rt_status_string", "rt_version"]
RTH_SYMBOLS = ["rth_app_create", "rth_app_destroy", "rth_app_finish", "rth_app_update", "rth_app_ctx",
               "rth_last_error", "rth_add_mesh", "rth_add_cuboid", "rth_add_plane", "rth_add_icosphere",
               "rth_add_material", "rth_spawn", "rth_set_transform", "rth_set_aabb", "rth_clear_world",
               "rth_spawn_default_scene", "rth_spawn_sphere_field", "rth_spawn_icosphere_soup", "rth_rotate_tagged",
               "rth_camera_look_at", "rth_camera_frame_scene", "rth_camera_frame_scene_at", "rth_camera_set_target", "rth_camera_set_projection",
               "rth_camera_set_graph", "rth_fly_cam_step", "rth_camera_view", "rth_prepare_meshinfo",
               "rth_info_counts", "rth_info_copy", "rth_log", "rth_entity_count", "rth_entity_get",
               "rth_fly_cam_run_script", "rth_write_ppm", "rth_write_png", "rth_update_counts"]

_lib = None


def lib():
    """The loaded librt_amd.so.  Raises if it has not been built: there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: run `make -C ray-tracing_amd` (or __graft_entry__.build()). "
                           "ray_tracing_amd has no CPU fallback.")
    # One HIP runtime per process: librt_amd.so asks for "libamdhip64.so", the name PyTorch-ROCm loads its bundled runtime
    # under, so with torch in the process first both share that runtime (torch streams and tensors can be handed to the
    # library); loaded the other way round, torch would bring a second runtime that may find no GPU.  So torch goes first
    # wherever it exists (tests, bench); without torch the library takes /opt/rocm/lib's runtime.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, u32, u64, sz, fp, i = C.c_void_p, C.c_uint32, C.c_uint64, C.c_size_t, C.POINTER(C.c_float), C.c_int
    sig = {
        "rt_create": (i, [i, C.POINTER(vp)]), "rt_destroy": (i, [vp]), "rt_set_stream": (i, [vp, vp]), "rt_set_frames_in_flight": (i, [vp, u32]), "rt_get_stream": (i, [vp, C.POINTER(vp)]),
        "rt_upload_scene": (i, [vp, vp, sz, vp, sz, vp, sz, vp, sz]),
        "rt_upload_scene_ex": (i, [vp, vp, sz, vp, sz, vp, sz, vp, sz, vp, vp, sz]),
        "rt_update_vertices": (i, [vp, vp, sz, vp, sz]),
        "rt_update_vertex_range": (i, [vp, vp, sz, sz, vp, sz, sz]),
        "rt_host_bvh_refit": (i, [vp, vp, sz, vp, sz, i]),
        "rt_set_view": (i, [vp, fp, fp, u32, u32, C.POINTER(u32)]),
        "rt_render": (i, [vp, C.POINTER(RenderParams)]), "rt_synchronize": (i, [vp]),
        "rt_read_radiance": (i, [vp, vp, sz]), "rt_read_prim_ids": (i, [vp, vp, sz]),
        "rt_read_hit_t": (i, [vp, vp, sz]), "rt_read_bgra8_srgb": (i, [vp, vp, sz]),
        "rt_read_accumulation": (i, [vp, vp, sz]),
        "rt_device_buffers": (i, [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]),
        "rt_resolve_accumulation": (i, [vp, u32]),
        "rt_encode_bgra8_srgb": (i, [vp, vp, sz, vp]),
        "rt_present_async": (i, [vp, vp, sz, vp]), "rt_present_wait": (i, [vp, C.c_uint32]),
        "rt_tile_slot_bytes": (i, [vp, u32, u32, C.POINTER(sz)]),
        "rt_pack_tiles": (i, [vp, u32, u32, u32, vp, sz]),
        "rt_unpack_tiles": (i, [vp, u32, u32, u32, u32, vp, sz]),
        "rt_render_static": (i, [vp, C.POINTER(RenderParams), vp, vp, vp, vp, sz]),
        "rt_get_counters": (i, [vp, C.POINTER(Counters)]), "rt_get_bvh_info": (i, [vp, C.POINTER(BvhInfo)]),
        "rt_read_bvh": (i, [vp, vp, sz, vp, sz]),
        "rt_host_bvh_build": (i, [vp, sz, vp, sz, vp, sz, i, C.POINTER(vp), C.POINTER(BvhInfo)]),
        "rt_host_bvh_copy": (i, [vp, vp, sz, vp, sz]), "rt_host_bvh_free": (None, [vp]),
        "rt_read_bvh_wide": (i, [vp, vp, sz]),
        "rt_host_bvh_wide": (i, [vp, vp, vp, sz, C.POINTER(sz), C.POINTER(C.c_uint32)]),
        "rt_last_error": (C.c_char_p, [vp]), "rt_status_string": (C.c_char_p, [i]), "rt_version": (C.c_char_p, []),
        "rt_measure_gather_rates": (i, [i, C.POINTER(C.c_double), sz]),
        "rth_app_create": (vp, []), "rth_app_destroy": (None, [vp]), "rth_app_finish": (i, [vp, i]),
        "rth_app_update": (i, [vp, C.POINTER(RenderParams)]), "rth_app_ctx": (vp, [vp]),
        "rth_last_error": (C.c_char_p, [vp]),
        "rth_add_mesh": (i, [vp, vp, vp, u32, vp, u32]), "rth_add_cuboid": (i, [vp, C.c_float, C.c_float, C.c_float]),
        "rth_add_plane": (i, [vp, C.c_float, C.c_float]), "rth_add_icosphere": (i, [vp, C.c_float, u32]),
        "rth_add_material": (i, [vp, fp]), "rth_spawn": (i, [vp, i, i, fp, fp, fp]),
        "rth_set_transform": (i, [vp, i, fp, fp, fp]), "rth_set_aabb": (i, [vp, i, fp, fp]),
        "rth_clear_world": (None, [vp]), "rth_spawn_default_scene": (None, [vp]),
        "rth_spawn_sphere_field": (None, [vp, u32, u64, u32]), "rth_spawn_icosphere_soup": (None, [vp, u32, u64, u32]),
        "rth_rotate_tagged": (None, [vp, C.c_float]), "rth_camera_look_at": (None, [vp, fp, fp, fp]),
        "rth_camera_frame_scene": (None, [vp]), "rth_camera_frame_scene_at": (None, [vp, C.c_float]), "rth_camera_set_target": (None, [vp, u32, u32, C.POINTER(u32)]),
        "rth_camera_set_projection": (None, [vp, C.c_float, C.c_float]), "rth_camera_set_graph": (None, [vp, i]),
        "rth_fly_cam_step": (None, [vp, C.c_float, C.c_float, u32, C.c_float]),
        "rth_fly_cam_run_script": (C.c_long, [vp, C.c_char_p]),
        "rth_update_counts": (None, [vp, C.POINTER(u32), C.POINTER(u32)]),
        "rth_write_ppm": (i, [C.c_char_p, vp, u32, u32]), "rth_write_png": (i, [C.c_char_p, vp, u32, u32]),
        "rth_camera_view": (None, [vp, fp, fp]), "rth_prepare_meshinfo": (i, [vp]),
        "rth_info_counts": (None, [vp, C.POINTER(sz), C.POINTER(sz), C.POINTER(sz), C.POINTER(sz)]),
        "rth_info_copy": (None, [vp, vp, vp, vp, vp]), "rth_log": (C.c_char_p, [vp]),
        "rth_entity_count": (sz, [vp]), "rth_entity_get": (i, [vp, sz, C.POINTER(EntityView)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    _lib = L
    return L


class RtError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__(f"rt status {status} ({lib().rt_status_string(status).decode()}): {msg}")
        self.status = status


def _f32(a, n=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if n is not None and a.size != n:
        raise ValueError(f"expected {n} floats, got {a.size}")
    return a


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class Scene:
    """Host arrays in the T1-T4 layouts (what RayTracingInfo holds)."""

    def __init__(self, triangles, meshes, vertices, materials, spheres=None, sphere_materials=None):
        self.triangles = np.ascontiguousarray(triangles, dtype=TRIANGLE)
        self.meshes = np.ascontiguousarray(meshes, dtype=MESH_INFO)
        self.vertices = np.ascontiguousarray(vertices, dtype=VERTEX)
        self.materials = np.ascontiguousarray(materials, dtype=MATERIAL)
        # extension: analytic spheres (rt_upload_scene_ex); sphere_materials[i] indexes `materials`
        self.spheres = np.ascontiguousarray(spheres if spheres is not None else np.zeros(0, SPHERE), dtype=SPHERE)
        self.sphere_materials = np.ascontiguousarray(
            sphere_materials if sphere_materials is not None else np.zeros(self.spheres.size, np.uint32), dtype=np.uint32)
        if self.sphere_materials.size != self.spheres.size:
            raise ValueError("one material index per sphere")


def scene_rs_default():
    """The default scene of the reference's orphaned `Scene` resource (src/scene.rs:22-52; dead code there: the module
    is commented out of src/lib.rs and its imports do not exist): three analytic spheres -- unit spheres at the origin and
    at (2, 0, -1), a radius-100 "ground" sphere at (0, -101, 0) -- with their albedo as the material colour, taken as
    written (no colour-space conversion; roughness and the three point lights of :63-67 have no counterpart in the
    live shader's SimpleMaterial / shading and are dropped).  Build-defined; renders through rt_upload_scene_ex."""
    spheres = np.zeros(3, SPHERE)
    spheres["center"] = [(0.0, 0.0, 0.0), (2.0, 0.0, -1.0), (0.0, -101.0, 0.0)]
    spheres["radius"] = [1.0, 1.0, 100.0]
    mats = np.zeros(3, MATERIAL)
    mats["color"] = [(1.0, 0.0, 1.0, 1.0), (0.2, 0.7, 0.1, 1.0), (0.2, 0.3, 6.0, 1.0)]
    return Scene(np.zeros(0, TRIANGLE), np.zeros(0, MESH_INFO), np.zeros(0, VERTEX), mats, spheres,
                 np.arange(3, dtype=np.uint32))


def analytic_sphere_field(n, seed):
    """SURVEY 8d config 2, variant (ii): the sphere field of App.spawn_sphere_field as ANALYTIC spheres
    (same SplitMix64 stream, so the same centres, radii and colours as the tessellated field) over the same
    ground plane mesh."""
    app = App()
    app.spawn_sphere_field(0, seed, 0)          # the ground plane and its material only
    ground = app.prepare_meshinfo()
    app.close()
    f32 = np.float32
    state = [int(seed) & 0xFFFFFFFFFFFFFFFF]

    def uniform(lo, hi):                        # SplitMix64 + 24-bit mantissa, as host/ray_tracing.cpp
        state[0] = (state[0] + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = state[0]
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        z ^= z >> 31
        u = f32(z >> 40) * f32(1.0 / 16777216.0)
        return f32(lo) + (f32(hi) - f32(lo)) * u

    spheres = np.zeros(n, SPHERE)
    mats = np.zeros(ground.materials.size + n, MATERIAL)
    mats[:ground.materials.size] = ground.materials
    for k in range(n):
        spheres["center"][k] = (uniform(-50.0, 50.0), uniform(0.0, 20.0), uniform(-50.0, 50.0))
        spheres["radius"][k] = uniform(0.2, 0.8)
        mats["color"][ground.materials.size + k] = (uniform(0.1, 1.0), uniform(0.1, 1.0), uniform(0.1, 1.0), 1.0)
    idx = np.arange(ground.materials.size, ground.materials.size + n, dtype=np.uint32)
    return Scene(ground.triangles, ground.meshes, ground.vertices, mats, spheres, idx)


GATHER_LEVELS = ("l1_16KiB", "l2_2MiB", "infinity_cache_128MiB", "hbm_1GiB", "hbm_4GiB")


def measure_gather_rates(device=0):
    """rt_measure_gather_rates: {level: {"independent": Ggather/s, "dependent": Ggather/s}} of 64-B gathers on this GPU."""
    vals = (C.c_double * 10)()
    rc = lib().rt_measure_gather_rates(device, vals, 10)
    if rc != RT_OK:
        raise RtError(rc, "rt_measure_gather_rates")
    return {name: {"independent": vals[2 * k], "dependent": vals[2 * k + 1]} for k, name in enumerate(GATHER_LEVELS)}


def write_image(path, bgra):
    """The presented frame (H, W, 4 uint8 BGRA, sRGB-encoded: Context.read_bgra8_srgb) as .ppm or .png."""
    bgra = np.ascontiguousarray(bgra, dtype=np.uint8)
    h, w = bgra.shape[:2]
    fn = lib().rth_write_png if str(path).lower().endswith(".png") else lib().rth_write_ppm
    rc = fn(str(path).encode(), bgra.ctypes.data, w, h)
    if rc != 0:
        raise OSError(f"write_image({path}): status {rc}")


def host_bvh_build(scene, threads=0, refit_to=None, wide=False):
    """BVH built on the host only (no GPU): returns (info, nodes, tri_records).
    refit_to: a Scene with the same topology and moved vertices; the tree is refitted to it before the copy.
    wide: also return the 4-wide quantised tree made from it: (info, nodes, recs, wide_nodes, wide_src, stack_need)."""
    h, info = C.c_void_p(), BvhInfo()
    t, m, v = scene.triangles, scene.meshes, scene.vertices
    rc = lib().rt_host_bvh_build(t.ctypes.data, t.size, m.ctypes.data, m.size, v.ctypes.data, v.size, threads,
                                 C.byref(h), C.byref(info))
    if rc != RT_OK:
        raise RtError(rc, lib().rt_last_error(None).decode())
    if refit_to is not None:
        t2, v2 = refit_to.triangles, refit_to.vertices
        rc = lib().rt_host_bvh_refit(h, t2.ctypes.data, t2.size, v2.ctypes.data, v2.size, threads)
        if rc != RT_OK:
            lib().rt_host_bvh_free(h)
            raise RtError(rc, "rt_host_bvh_refit")
    nodes, recs = np.empty(info.n_nodes, dtype=BVH_NODE), np.empty(info.n_triangles, dtype=TRI_RECORD)
    lib().rt_host_bvh_copy(h, nodes.ctypes.data, nodes.nbytes, recs.ctypes.data, recs.nbytes)
    if wide:
        n, need = C.c_size_t(), C.c_uint32()
        lib().rt_host_bvh_wide(h, None, None, 0, C.byref(n), C.byref(need))
        wn, ws = np.zeros(n.value, dtype=WIDE_NODE), np.zeros(n.value, dtype=WIDE_SRC)
        rc = lib().rt_host_bvh_wide(h, wn.ctypes.data, ws.ctypes.data, n.value, C.byref(n), C.byref(need))
        lib().rt_host_bvh_free(h)
        if rc != RT_OK:
            raise RtError(rc, "rt_host_bvh_wide")
        return info, nodes, recs, wn, ws, need.value
    lib().rt_host_bvh_free(h)
    return info, nodes, recs


class Context:
    """One rt_ctx: what RayTracingPipeline + RayTracingPassNode hold on the GPU side."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        rc = lib().rt_create(device, C.byref(self._h))
        if rc != RT_OK:
            raise RtError(rc, lib().rt_last_error(None).decode())
        self.width = self.height = 0

    def close(self):
        if self._h:
            lib().rt_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, allow_not_ready=False):
        if rc == RT_OK or (allow_not_ready and rc == RT_NOT_READY):
            return rc
        raise RtError(rc, lib().rt_last_error(self._h).decode())

    def set_stream(self, stream_ptr):
        self._chk(lib().rt_set_stream(self._h, C.c_void_p(stream_ptr)))

    def get_stream(self):
        """hipStream_t (int; 0 = the null stream) the frame rendered last runs on."""
        p = C.c_void_p()
        self._chk(lib().rt_get_stream(self._h, C.byref(p)))
        return p.value or 0

    def set_frames_in_flight(self, n):
        self._chk(lib().rt_set_frames_in_flight(self._h, n))

    def upload_scene(self, scene):
        t, m, v, c = scene.triangles, scene.meshes, scene.vertices, scene.materials
        sp, sm = getattr(scene, "spheres", None), getattr(scene, "sphere_materials", None)
        if sp is not None and sp.size:
            return self._chk(lib().rt_upload_scene_ex(self._h, t.ctypes.data, t.size, m.ctypes.data, m.size,
                                                      v.ctypes.data, v.size, c.ctypes.data, c.size,
                                                      sp.ctypes.data, sm.ctypes.data, sp.size))
        return self._chk(lib().rt_upload_scene(self._h, t.ctypes.data, t.size, m.ctypes.data, m.size,
                                               v.ctypes.data, v.size, c.ctypes.data, c.size))

    def update_vertices(self, scene):
        v, m = scene.vertices, scene.meshes
        return self._chk(lib().rt_update_vertices(self._h, v.ctypes.data, v.size, m.ctypes.data, m.size))

    def update_vertex_range(self, scene, first_vertex, n_vertices, first_mesh, n_meshes):
        """Only meshes [first_mesh, first_mesh + n_meshes) of `scene` moved; their vertices are [first_vertex, +n_vertices)."""
        v = np.ascontiguousarray(scene.vertices[first_vertex:first_vertex + n_vertices])
        m = np.ascontiguousarray(scene.meshes[first_mesh:first_mesh + n_meshes])
        return self._chk(lib().rt_update_vertex_range(self._h, v.ctypes.data, first_vertex, v.size, m.ctypes.data, first_mesh, m.size))

    def set_view(self, world_from_clip, world_position, width, height, viewport=None):
        m, p = _f32(world_from_clip, 16), _f32(world_position, 3)
        vp = (C.c_uint32 * 4)(*viewport) if viewport is not None else None
        self._chk(lib().rt_set_view(self._h, _fp(m), _fp(p), width, height, vp))
        self.width, self.height = width, height

    def render(self, flags=0, spp=0, sample_first=0, max_segments=0, seed=0, stream_id=0, shard_rank=0,
               shard_count=0, tile_size=0, frame_count=0):
        p = RenderParams(C.sizeof(RenderParams), flags, frame_count, spp, sample_first, max_segments, seed, stream_id,
                         shard_rank, shard_count, tile_size, 0)
        return self._chk(lib().rt_render(self._h, C.byref(p)), allow_not_ready=True)

    def render_static(self, d_radiance, d_prim, d_hit_t, d_accum, n_pixels, flags=0, spp=0, sample_first=0, max_segments=0,
                      seed=0, stream_id=0, shard_rank=0, shard_count=0, tile_size=0, frame_count=0):
        """rt_render_static: the same frame into the CALLER's device buffers (raw device pointers), allocating nothing."""
        p = RenderParams(C.sizeof(RenderParams), flags, frame_count, spp, sample_first, max_segments, seed, stream_id,
                         shard_rank, shard_count, tile_size, 0)
        return self._chk(lib().rt_render_static(self._h, C.byref(p), C.c_void_p(d_radiance), C.c_void_p(d_prim),
                                                C.c_void_p(d_hit_t), C.c_void_p(d_accum), n_pixels), allow_not_ready=True)

    def synchronize(self):
        self._chk(lib().rt_synchronize(self._h))

    def encode_bgra8_srgb(self, rgba):
        """Any (..., 4) float32 host array through the device's Bgra8UnormSrgb encode (k_encode_bgra8)."""
        rgba = np.ascontiguousarray(rgba, dtype=np.float32)
        out = np.empty(rgba.shape, dtype=np.uint8)
        self._chk(lib().rt_encode_bgra8_srgb(self._h, rgba.ctypes.data, rgba.size // 4, out.ctypes.data))
        return out

    def tile_slot_bytes(self, shard_count, tile_size=64):
        n = C.c_size_t()
        self._chk(lib().rt_tile_slot_bytes(self._h, shard_count, tile_size, C.byref(n)))
        return n.value

    def pack_tiles(self, shard_rank, shard_count, tile_size, d_packed, n_bytes):
        """This rank's shard tiles of the radiance buffer, packed, into the device buffer d_packed (raw pointer)."""
        self._chk(lib().rt_pack_tiles(self._h, shard_rank, shard_count, tile_size, C.c_void_p(d_packed), n_bytes))

    def unpack_tiles(self, first_rank, n_ranks, shard_count, tile_size, d_packed, slot_bytes):
        """Packed tiles of ranks [first_rank, first_rank + n_ranks), slot_bytes apart in d_packed, into the radiance buffer."""
        self._chk(lib().rt_unpack_tiles(self._h, first_rank, n_ranks, shard_count, tile_size, C.c_void_p(d_packed), slot_bytes))

    def _read(self, fn, dtype, shape):
        out = np.empty(shape, dtype=dtype)
        self._chk(fn(self._h, out.ctypes.data, out.size))
        return out

    def read_radiance(self):
        return self._read(lib().rt_read_radiance, np.float32, (self.height, self.width, 4))

    def read_accumulation(self):
        return self._read(lib().rt_read_accumulation, np.float32, (self.height, self.width, 4))

    def read_prim_ids(self):
        return self._read(lib().rt_read_prim_ids, np.uint32, (self.height, self.width))

    def read_hit_t(self):
        return self._read(lib().rt_read_hit_t, np.float32, (self.height, self.width))

    def read_bgra8_srgb(self):
        return self._read(lib().rt_read_bgra8_srgb, np.uint8, (self.height, self.width, 4))

    def present_async(self, host_bgra):
        """rt_present_async: encode + copy of the frame rendered last into `host_bgra` (uint8, H*W*4, C-contiguous; the
        caller keeps it alive and untouched until present_wait(ticket)), on that frame's stream; returns the ticket."""
        assert host_bgra.dtype == np.uint8 and host_bgra.flags["C_CONTIGUOUS"]
        t = C.c_uint32()
        self._chk(lib().rt_present_async(self._h, host_bgra.ctypes.data, host_bgra.size, C.byref(t)))
        return t.value

    def present_wait(self, ticket):
        self._chk(lib().rt_present_wait(self._h, ticket))

    def device_buffers(self):
        r, p, a = C.c_void_p(), C.c_void_p(), C.c_void_p()
        self._chk(lib().rt_device_buffers(self._h, C.byref(r), C.byref(p), C.byref(a)))
        return r.value, p.value, a.value

    def resolve_accumulation(self, total_samples):
        self._chk(lib().rt_resolve_accumulation(self._h, total_samples))

    def counters(self):
        c = Counters()
        self._chk(lib().rt_get_counters(self._h, C.byref(c)))
        return c

    def bvh_info(self):
        b = BvhInfo()
        self._chk(lib().rt_get_bvh_info(self._h, C.byref(b)))
        return b

    def read_bvh(self):
        b = self.bvh_info()
        nodes = np.empty(b.n_nodes, dtype=BVH_NODE)
        recs = np.empty(b.n_triangles, dtype=TRI_RECORD)
        self._chk(lib().rt_read_bvh(self._h, nodes.ctypes.data, nodes.nbytes, recs.ctypes.data, recs.nbytes))
        return nodes, recs

    def read_bvh_wide(self):
        wide = np.empty(self.bvh_info().n_wide_nodes, dtype=WIDE_NODE)
        self._chk(lib().rt_read_bvh_wide(self._h, wide.ctypes.data, wide.nbytes))
        return wide


class _BorrowedContext(Context):
    """The rt_ctx owned by an App's RayTracingPipeline (not destroyed here)."""

    def __init__(self, handle, app):
        self._h = C.c_void_p(handle)
        self._app = app
        self.width = self.height = 0

    def close(self):
        self._h = C.c_void_p()


class App:
    """App + World + RayTracingPlugin (src/ray_tracing.rs:26-49, src/main.rs:13-26)."""

    def __init__(self):
        self._h = C.c_void_p(lib().rth_app_create())
        if not self._h:
            raise MemoryError("rth_app_create failed")
        self._ctx = None
        self.width, self.height = 1280, 720

    def close(self):
        if self._h:
            lib().rth_app_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def err(self):
        return lib().rth_last_error(self._h).decode()

    # --- world
    def add_mesh(self, positions, normals, indices):
        p = None if positions is None else _f32(positions)
        n = None if normals is None else _f32(normals)
        ix = None if indices is None else np.ascontiguousarray(indices, dtype=np.uint32)
        nv = (p.size if p is not None else (n.size if n is not None else 0)) // 3
        return lib().rth_add_mesh(self._h, p.ctypes.data if p is not None else None,
                                  n.ctypes.data if n is not None else None, nv,
                                  ix.ctypes.data if ix is not None else None, ix.size if ix is not None else 0)

    def add_cuboid(self, sx, sy, sz):
        return lib().rth_add_cuboid(self._h, sx, sy, sz)

    def add_plane(self, hx, hz):
        return lib().rth_add_plane(self._h, hx, hz)

    def add_icosphere(self, radius, subdivisions=5):
        return lib().rth_add_icosphere(self._h, radius, subdivisions)

    def add_material(self, rgba):
        return lib().rth_add_material(self._h, _fp(_f32(rgba, 4)))

    def spawn(self, mesh, material, scale=(1, 1, 1), rotation=(0, 0, 0, 1), translation=(0, 0, 0)):
        return lib().rth_spawn(self._h, mesh, material, _fp(_f32(scale, 3)), _fp(_f32(rotation, 4)),
                               _fp(_f32(translation, 3)))

    def set_transform(self, entity, scale=(1, 1, 1), rotation=(0, 0, 0, 1), translation=(0, 0, 0)):
        return lib().rth_set_transform(self._h, entity, _fp(_f32(scale, 3)), _fp(_f32(rotation, 4)),
                                       _fp(_f32(translation, 3)))

    def set_aabb(self, entity, center, half_extents):
        return lib().rth_set_aabb(self._h, entity, _fp(_f32(center, 3)), _fp(_f32(half_extents, 3)))

    def clear_world(self):
        lib().rth_clear_world(self._h)

    def spawn_default_scene(self):
        lib().rth_spawn_default_scene(self._h)

    def spawn_sphere_field(self, n, seed, subdivisions=5):
        lib().rth_spawn_sphere_field(self._h, n, seed, subdivisions)

    def spawn_icosphere_soup(self, n, seed, subdivisions=5):
        lib().rth_spawn_icosphere_soup(self._h, n, seed, subdivisions)

    def rotate_tagged(self, dt):
        lib().rth_rotate_tagged(self._h, dt)

    # --- camera
    def camera_look_at(self, eye, target=(0, 0, 0), up=(0, 1, 0)):
        lib().rth_camera_look_at(self._h, _fp(_f32(eye, 3)), _fp(_f32(target, 3)), _fp(_f32(up, 3)))

    def camera_frame_scene(self, radii=None):
        if radii is None:
            lib().rth_camera_frame_scene(self._h)
        else:
            lib().rth_camera_frame_scene_at(self._h, radii)

    def camera_set_target(self, width, height, viewport=None):
        vp = (C.c_uint32 * 4)(*viewport) if viewport is not None else None
        lib().rth_camera_set_target(self._h, width, height, vp)
        self.width, self.height = width, height

    def camera_set_projection(self, fov_y, near):
        lib().rth_camera_set_projection(self._h, fov_y, near)

    def camera_set_graph(self, ray_tracing):
        lib().rth_camera_set_graph(self._h, 1 if ray_tracing else 0)

    def fly_cam_step(self, mouse_dx=0.0, mouse_dy=0.0, keys=0, dt=0.0):
        lib().rth_fly_cam_step(self._h, mouse_dx, mouse_dy, keys, dt)

    def update_counts(self):
        """(ranged, full) vertex-only updates RayTracingPassNode::run has made."""
        a, b = C.c_uint32(), C.c_uint32()
        lib().rth_update_counts(self._h, C.byref(a), C.byref(b))
        return a.value, b.value

    def fly_cam_run_script(self, text):
        """One fly-cam step per line: 'dt mouse_dx mouse_dy [keys]' (rt_host.h). Returns the number of steps."""
        n = lib().rth_fly_cam_run_script(self._h, text.encode())
        if n < 0:
            raise ValueError(f"fly-cam script: malformed line {-n}")
        return n

    def camera_view(self):
        m, p = np.zeros(16, np.float32), np.zeros(3, np.float32)
        lib().rth_camera_view(self._h, _fp(m), _fp(p))
        return m, p

    # --- extraction
    def prepare_meshinfo(self):
        if lib().rth_prepare_meshinfo(self._h) != 0:
            raise RuntimeError(self.err())
        return self.info()

    def info(self):
        nt, nm, nv, nc = C.c_size_t(), C.c_size_t(), C.c_size_t(), C.c_size_t()
        lib().rth_info_counts(self._h, C.byref(nt), C.byref(nm), C.byref(nv), C.byref(nc))
        t, m = np.zeros(nt.value, TRIANGLE), np.zeros(nm.value, MESH_INFO)
        v, c = np.zeros(nv.value, VERTEX), np.zeros(nc.value, MATERIAL)
        lib().rth_info_copy(self._h, t.ctypes.data, m.ctypes.data, v.ctypes.data, c.ctypes.data)
        return Scene(t, m, v, c)

    def log(self):
        return lib().rth_log(self._h).decode()

    def entities(self):
        out = []
        for k in range(lib().rth_entity_count(self._h)):
            e = EntityView()
            lib().rth_entity_get(self._h, k, C.byref(e))
            nv, ni = e.n_vertices, e.n_indices
            out.append(dict(
                positions=np.ctypeslib.as_array(e.positions, (nv, 3)).copy() if e.positions else None,
                normals=np.ctypeslib.as_array(e.normals, (nv, 3)).copy() if e.normals else None,
                indices=np.ctypeslib.as_array(e.indices, (ni,)).copy() if e.indices else None,
                base_color_linear=np.array(e.base_color_linear, np.float32), scale=np.array(e.scale, np.float32),
                rotation=np.array(e.rotation, np.float32), translation=np.array(e.translation, np.float32),
                aabb_center=np.array(e.aabb_center, np.float32),
                aabb_half_extents=np.array(e.aabb_half_extents, np.float32)))
        return out

    # --- plugin lifecycle
    def finish(self, device=0):
        rc = lib().rth_app_finish(self._h, device)
        if rc != RT_OK:
            raise RtError(rc, self.err())
        return rc

    def update(self, **kw):
        flags = kw.pop("flags", 0)
        p = RenderParams(C.sizeof(RenderParams), flags, 0, kw.pop("spp", 0), kw.pop("sample_first", 0),
                         kw.pop("max_segments", 0), kw.pop("seed", 0), kw.pop("stream_id", 0),
                         kw.pop("shard_rank", 0), kw.pop("shard_count", 0), kw.pop("tile_size", 0), 0)
        if kw:
            raise TypeError(f"unknown arguments {list(kw)}")
        rc = lib().rth_app_update(self._h, C.byref(p))
        if rc < 0:
            raise RtError(rc, self.err())
        return rc

    def ctx(self):
        h = lib().rth_app_ctx(self._h)
        if not h:
            return None
        c = _BorrowedContext(h, self)
        c.width, c.height = self.width, self.height
        return c
