//! UNBUILT SOURCE: `extern "C"` view of include/rt_amd.h for camms205/ray-tracing's src/node.rs and
//! src/pipeline.rs.  Mechanical from the header; never compiled here (no Rust toolchain in the image).
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)]
pub struct rt_ctx { _private: [u8; 0] }

/// T1 Triangle, src/ray_tracing.rs:51-54 (12 B)
#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct rt_triangle { pub indices: [u32; 3] }
/// T2 MeshInfo, src/ray_tracing.rs:56-63 in its std430 layout (48 B)
#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct rt_mesh_info {
    pub first_tri: u32, pub tri_count: u32, pub material: u32, pub _pad0: u32,
    pub aabb_min: [f32; 3], pub _pad1: f32,
    pub aabb_max: [f32; 3], pub _pad2: f32,
}
/// T3 vertex, src/ray_tracing.rs:92-93 ([Vec3; 2] with 16-B stride, 32 B)
#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct rt_vertex { pub pos: [f32; 3], pub _pad0: f32, pub norm: [f32; 3], pub _pad1: f32 }
/// T4 SimpleMaterial, src/ray_tracing.rs:65-68 (16 B)
#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct rt_material { pub color: [f32; 4] }
#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct rt_sphere { pub center: [f32; 3], pub radius: f32 }

#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct rt_render_params {
    pub struct_size: u32, pub flags: u32, pub frame_count: u32, pub spp: u32, pub sample_first: u32,
    pub max_segments: u32, pub seed: u32, pub stream_id: u32, pub shard_rank: u32, pub shard_count: u32,
    pub tile_size: u32, pub _reserved: u32,
}

pub const RT_OK: c_int = 0;
pub const RT_NOT_READY: c_int = 1;
pub const RT_ERR_ARG: c_int = -1;
pub const RT_ERR_HIP: c_int = -2;
pub const RT_ERR_NO_GPU: c_int = -3;
pub const RT_ERR_SCENE: c_int = -4;
pub const RT_ERR_OOM: c_int = -5;

pub const RT_FLAG_SHADOW: u32 = 0x0001;
pub const RT_FLAG_NO_MESH_CULL: u32 = 0x0002;
pub const RT_FLAG_BRUTE_FORCE: u32 = 0x0004;
pub const RT_FLAG_COUNTERS: u32 = 0x0008;
pub const RT_FLAG_PATH_TRACE: u32 = 0x0010;
pub const RT_FLAG_ACCUMULATE: u32 = 0x0020;
pub const RT_FLAG_TIME_KERNELS: u32 = 0x0040;
pub const RT_FLAG_WAVE_TRAVERSAL: u32 = 0x0080;
pub const RT_FLAG_STATIC_GRID: u32 = 0x0100;
pub const RT_FLAG_BINARY_TREE: u32 = 0x0200;

/// Counters of the last `rt_render` (include/rt_amd.h `rt_counters`).
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct rt_counters {
    pub rays_primary: u64, pub rays_shadow: u64, pub rays_bounce: u64,
    pub node_visits: u64, pub tri_tests: u64, pub mesh_aabb_tests: u64,
    pub kernel_ms: f64, pub kernel_launches: u32, pub _pad: u32,
    pub wave_node_iters: u64, pub wave_tri_iters: u64, pub wave_alive_lanes: u64,
}
/// What the library built over the scene (include/rt_amd.h `rt_bvh_info`).
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct rt_bvh_info {
    pub n_nodes: u64, pub n_triangles: u64, pub n_meshes: u64, pub n_vertices: u64,
    pub node_bytes: u32, pub tri_bytes: u32, pub max_depth: u32, pub max_leaf: u32,
    pub build_ms: f64, pub sah_cost: f64, pub n_builds: u32, pub n_refits: u32,
    pub n_wide_nodes: u64, pub wide_max_depth: u32, pub wide_stack_need: u32,
}
/// Opaque host-only tree (no GPU touched): `rt_host_bvh_*`.
#[repr(C)]
pub struct rt_host_bvh { _private: [u8; 0] }

extern "C" {
    pub fn rt_get_counters(ctx: *mut rt_ctx, out: *mut rt_counters) -> c_int;
    pub fn rt_get_bvh_info(ctx: *mut rt_ctx, out: *mut rt_bvh_info) -> c_int;
    pub fn rt_read_bvh(ctx: *mut rt_ctx, nodes: *mut c_void, node_bytes_total: usize, tris: *mut c_void, tri_bytes_total: usize) -> c_int;
    pub fn rt_read_bvh_wide(ctx: *mut rt_ctx, wide_nodes: *mut c_void, bytes_total: usize) -> c_int;
    pub fn rt_host_bvh_build(triangles: *const rt_triangle, n_triangles: usize, meshes: *const rt_mesh_info, n_meshes: usize,
        vertices: *const rt_vertex, n_vertices: usize, n_threads: c_int, out: *mut *mut rt_host_bvh, info: *mut rt_bvh_info) -> c_int;
    pub fn rt_host_bvh_refit(bvh: *mut rt_host_bvh, triangles: *const rt_triangle, n_triangles: usize,
        vertices: *const rt_vertex, n_vertices: usize, n_threads: c_int) -> c_int;
    pub fn rt_host_bvh_copy(bvh: *const rt_host_bvh, nodes: *mut c_void, node_bytes_total: usize, tris: *mut c_void, tri_bytes_total: usize) -> c_int;
    pub fn rt_host_bvh_wide(bvh: *const rt_host_bvh, wide_nodes: *mut c_void, wide_src: *mut c_void, capacity_nodes: usize,
        n_wide: *mut usize, stack_need: *mut u32) -> c_int;
    pub fn rt_host_bvh_free(bvh: *mut rt_host_bvh);
    pub fn rt_create(device_ordinal: c_int, out: *mut *mut rt_ctx) -> c_int;
    pub fn rt_destroy(ctx: *mut rt_ctx) -> c_int;
    pub fn rt_set_stream(ctx: *mut rt_ctx, hip_stream: *mut c_void) -> c_int;
    /// 1..4 frames in flight inside one context (sets of frame buffers + streams taken in turn by rt_render).
    pub fn rt_set_frames_in_flight(ctx: *mut rt_ctx, n: u32) -> c_int;
    /// The stream (hipStream_t) of the frame rendered last, to order consumers of its buffers against.
    pub fn rt_get_stream(ctx: *mut rt_ctx, hip_stream: *mut *mut c_void) -> c_int;
    pub fn rt_upload_scene(ctx: *mut rt_ctx,
        triangles: *const rt_triangle, n_triangles: usize, meshes: *const rt_mesh_info, n_meshes: usize,
        vertices: *const rt_vertex, n_vertices: usize, materials: *const rt_material, n_materials: usize) -> c_int;
    /// Extension: the same plus analytic spheres (centre + radius as `Sphere` in src/scene.rs:26-28).
    pub fn rt_upload_scene_ex(ctx: *mut rt_ctx,
        triangles: *const rt_triangle, n_triangles: usize, meshes: *const rt_mesh_info, n_meshes: usize,
        vertices: *const rt_vertex, n_vertices: usize, materials: *const rt_material, n_materials: usize,
        spheres: *const rt_sphere, sphere_materials: *const u32, n_spheres: usize) -> c_int;
    pub fn rt_update_vertices(ctx: *mut rt_ctx, vertices: *const rt_vertex, n_vertices: usize,
        meshes: *const rt_mesh_info, n_meshes: usize) -> c_int;
    pub fn rt_update_vertex_range(ctx: *mut rt_ctx, vertices: *const rt_vertex, first_vertex: usize, n_vertices: usize,
        meshes: *const rt_mesh_info, first_mesh: usize, n_meshes: usize) -> c_int;
    pub fn rt_set_view(ctx: *mut rt_ctx, world_from_clip: *const f32, world_position: *const f32,
        width: u32, height: u32, viewport: *const u32) -> c_int;
    pub fn rt_render(ctx: *mut rt_ctx, params: *const rt_render_params) -> c_int;
    pub fn rt_synchronize(ctx: *mut rt_ctx) -> c_int;
    /// The same frame into the caller's device buffers, allocating nothing (scratch in a static 32 MiB device array;
    /// path tracing up to 322,560 pixels; calls that overlap on one GPU race).
    pub fn rt_render_static(ctx: *mut rt_ctx, params: *const rt_render_params, d_radiance: *mut c_void, d_prim_ids: *mut c_void,
        d_hit_t: *mut c_void, d_accum: *mut c_void, n_pixels: usize) -> c_int;
    pub fn rt_read_radiance(ctx: *mut rt_ctx, rgba: *mut f32, n_floats: usize) -> c_int;
    pub fn rt_read_prim_ids(ctx: *mut rt_ctx, prim: *mut u32, n: usize) -> c_int;
    pub fn rt_read_hit_t(ctx: *mut rt_ctx, t: *mut f32, n: usize) -> c_int;
    pub fn rt_read_bgra8_srgb(ctx: *mut rt_ctx, bgra: *mut u8, n_bytes: usize) -> c_int;
    pub fn rt_read_accumulation(ctx: *mut rt_ctx, accum: *mut f32, n_floats: usize) -> c_int;
    pub fn rt_device_buffers(ctx: *mut rt_ctx, d_radiance: *mut *mut c_void, d_prim: *mut *mut c_void,
        d_accum: *mut *mut c_void) -> c_int;
    pub fn rt_resolve_accumulation(ctx: *mut rt_ctx, total_samples: u32) -> c_int;
    /// Bgra8UnormSrgb encode of any host RGBA f32 frame through the device kernel (exact bytes).
    pub fn rt_encode_bgra8_srgb(ctx: *mut rt_ctx, rgba: *const f32, n_pixels: usize, bgra: *mut u8) -> c_int;
    /// Presentation that does not stall frames in flight: encode + copy of the frame rendered last on its own stream;
    /// `ticket` names the frame's set, `rt_present_wait` blocks until the bytes have arrived in `host_bgra`.
    pub fn rt_present_async(ctx: *mut rt_ctx, host_bgra: *mut u8, n_bytes: usize, ticket: *mut u32) -> c_int;
    pub fn rt_present_wait(ctx: *mut rt_ctx, ticket: u32) -> c_int;
    /// Multi-GPU tile exchange: packed size of the largest rank's tiles, pack own tiles, scatter peers' tiles.
    pub fn rt_tile_slot_bytes(ctx: *mut rt_ctx, shard_count: u32, tile_size: u32, bytes: *mut usize) -> c_int;
    pub fn rt_pack_tiles(ctx: *mut rt_ctx, shard_rank: u32, shard_count: u32, tile_size: u32, d_packed: *mut c_void, bytes: usize) -> c_int;
    pub fn rt_unpack_tiles(ctx: *mut rt_ctx, first_rank: u32, n_ranks: u32, shard_count: u32, tile_size: u32,
        d_packed: *const c_void, slot_bytes: usize) -> c_int;
    /// Measurement aid: 64-B gather rates of this GPU per level of the memory hierarchy (include/rt_amd.h); 10 values.
    pub fn rt_measure_gather_rates(device_ordinal: c_int, ggather_per_s: *mut f64, n_values: usize) -> c_int;
    pub fn rt_last_error(ctx: *mut rt_ctx) -> *const c_char;
    pub fn rt_status_string(status: c_int) -> *const c_char;
    pub fn rt_version() -> *const c_char;
}
