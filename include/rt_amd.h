/*
 * rt_amd.h -- C ABI of the MI355X-native per-pixel ray-tracing hot path.
 *
 * Drop-in boundary for the L1<->L0 seam of camms205/ray-tracing: what src/node.rs and
 * src/pipeline.rs hand to wgpu today.  The reference has no FFI of its own (SURVEY.md
 * section 8b); each entry point cites the reference interface it replaces.  Plain
 * pointers and sizes only; no C++/torch types; every call returns an int status; nothing
 * throws or aborts across this boundary.  INTEGRATION.md shows the Rust `-sys` binding.
 *
 * Threading: a context is not re-entrant; callers serialise per context (the reference
 * calls RayTracingPassNode::run from one render thread, src/node.rs:29).  One context per
 * GPU.  Host pointers are borrowed for the duration of the call only.
 */
#ifndef RT_AMD_H
#define RT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes ---------------------------------------------------------------- */
#define RT_OK            0
#define RT_NOT_READY     1   /* no scene or no view yet: the frame is skipped, mirroring the
                                silent `return Ok(())` of src/node.rs:52-56. Not an error. */
#define RT_ERR_ARG      -1
#define RT_ERR_HIP      -2   /* a HIP runtime call failed; rt_last_error() has the text */
#define RT_ERR_NO_GPU   -3   /* no usable gfx950 device: the library has NO CPU fallback */
#define RT_ERR_SCENE    -4   /* scene arrays inconsistent (index out of range, ...) */
#define RT_ERR_OOM      -5

/* ---- scene data contract: byte-identical to the RayTracingInfo bindings ---------- */
/* T1  Triangle   src/ray_tracing.rs:51-54, ray_tracing.wgsl:19-23.  12 B. */
typedef struct { uint32_t indices[3]; } rt_triangle;
/* T2  MeshInfo   src/ray_tracing.rs:56-63, ray_tracing.wgsl:24-30.  48 B (std430). */
typedef struct {
    uint32_t first_tri, tri_count, material, _pad0;
    float aabb_min[3]; float _pad1;
    float aabb_max[3]; float _pad2;
} rt_mesh_info;
/* T3  Vertex     src/ray_tracing.rs:92-93 ([Vec3;2]), ray_tracing.wgsl:14-17.  32 B. */
typedef struct { float pos[3]; float _pad0; float norm[3]; float _pad1; } rt_vertex;
/* T4  SimpleMaterial  src/ray_tracing.rs:65-68, ray_tracing.wgsl:32-34.  16 B. */
typedef struct { float color[4]; } rt_material;

/* EXTENSION (SURVEY 8d config 2 variant ii): analytic sphere, centre + radius as the dead Sphere type of
 * src/scene.rs:26-28 has them.  The reference has no ray/sphere test (its live spheres are tessellated,
 * src/main.rs:85): the test is build-defined (oracle/rt_oracle.h, orc_ray_sphere).  16 B. */
typedef struct { float center[3]; float radius; } rt_sphere;

typedef struct rt_ctx rt_ctx;

/* ---- lifecycle: replaces RayTracingPipeline::from_world (src/pipeline.rs:25-72) ----- */
int rt_create(int device_ordinal, rt_ctx **out);
int rt_destroy(rt_ctx *ctx);
/* Run all work of this context on an existing HIP stream (hipStream_t passed as void*);
 * NULL restores the context's own stream (created hipStreamNonBlocking: it does NOT order against
 * HIP's null stream); RT_STREAM_NULL selects HIP's null stream itself, explicitly. */
#define RT_STREAM_NULL ((void *)(intptr_t)-1)
int rt_set_stream(rt_ctx *ctx, void *hip_stream);
/* Frames in flight inside ONE context (1..4, default 1).  With n > 1 the context keeps n sets of frame buffers and n
 * streams and rt_render takes them in turn without waiting for the frame before: while the last 8x8 regions of frame k
 * drain, the waves of frame k + 1 already fill the compute units they leave idle (+20 % frames per second at 1080p on
 * the 7.2 M-triangle scene, one scene copy instead of the two that two contexts need).  The reference's render node
 * (src/node.rs:29-79) submits one frame after the other the same way and lets the GPU queue overlap them.
 * rt_read_*, rt_device_buffers, rt_pack_tiles and rt_get_counters refer to the frame rendered LAST (they wait for it
 * only); rt_synchronize waits for all.  Scene updates wait for every frame in flight.  Path-traced frames alternate
 * between two sets and share ONE accumulation buffer: a frame's tracing overlaps with the frame before it, its
 * additions to the running sums (and a restart's clear) wait for that frame's end, so sums and RT_FLAG_ACCUMULATE
 * chains are those of frames rendered one after the other; rt_read_accumulation and rt_resolve_accumulation wait for
 * them.  On a caller's stream (rt_set_stream) all sets run on that one stream and frames serialise.  Costs 24 B per
 * pixel per extra set, and a second set of path queues (104 B per path of a batch) once path-traced frames overlap.
 * With n > 1 every set renders on a stream that has a hardware queue of its own (created with a CU mask of all
 * compute units: ordinary streams share a few queues as the runtime sees fit, and two sets on one queue run their
 * frames strictly one after the other); such streams are of HIP's default kind, so unlike the context's ordinary
 * stream they also order against HIP's null stream. */
int rt_set_frames_in_flight(rt_ctx *ctx, uint32_t n);
/* The HIP stream (hipStream_t as void*) the frame rendered last runs on: the stream to order a consumer of
 * rt_device_buffers / rt_pack_tiles against (hipStreamWaitEvent, or make it the current stream of a framework), e.g. a
 * collective that sends frame k while frame k + 1 renders on the next set's stream. */
int rt_get_stream(rt_ctx *ctx, void **hip_stream);

/* ---- scene upload: replaces as_bind_group's per-frame buffer creation ---------------
 * (src/node.rs:42-50; bindings 1-4 of RayTracingInfo, src/ray_tracing.rs:84-96).
 * Host arrays in the exact T1-T4 layouts.  The library copies them, re-lays them out and
 * builds its BVH.  Call when the scene changes, not per frame. */
int rt_upload_scene(rt_ctx *ctx,
                    const rt_triangle *triangles, size_t n_triangles,
                    const rt_mesh_info *meshes, size_t n_meshes,
                    const rt_vertex *vertices, size_t n_vertices,
                    const rt_material *materials, size_t n_materials);

/* EXTENSION: the same, plus analytic spheres.  Spheres come after every mesh in the reference's loop
 * order (ties on t go to the triangle, then to the lower sphere index); a sphere's primitive id is
 * n_triangles + its index; sphere_materials[i] indexes `materials`.  Only the entry point of a ray counts
 * (an origin inside a sphere misses it, as back faces do for triangles, wgsl:137). */
int rt_upload_scene_ex(rt_ctx *ctx,
                       const rt_triangle *triangles, size_t n_triangles,
                       const rt_mesh_info *meshes, size_t n_meshes,
                       const rt_vertex *vertices, size_t n_vertices,
                       const rt_material *materials, size_t n_materials,
                       const rt_sphere *spheres, const uint32_t *sphere_materials, size_t n_spheres);

/* Same scene, moved vertices (the reference rebuilds and re-uploads everything every
 * frame, src/ray_tracing.rs:113-168, and main.rs rotates the cube every frame, src/main.rs:122-126).
 * Triangles, mesh ranges and materials must be the ones last uploaded; `vertices` replaces T3 and
 * `meshes` T2 (their AABBs move with the vertices).  Leaf records are re-baked and the BVH boxes refitted
 * instead of rebuilt.  Returns RT_ERR_SCENE if the topology differs (call rt_upload_scene instead). */
int rt_update_vertices(rt_ctx *ctx, const rt_vertex *vertices, size_t n_vertices,
                       const rt_mesh_info *meshes, size_t n_meshes);
/* The dirty-mesh form of the same ("re-bake only moved meshes"): only meshes
 * [first_mesh, first_mesh + n_meshes) moved.  `meshes` holds their n_meshes new T2 entries, `vertices` the
 * n_vertices new T3 entries that replace [first_vertex, first_vertex + n_vertices) -- a range that must cover
 * every vertex those meshes' triangles use and none that a triangle of another mesh uses.  Only these bytes
 * cross PCIe; on the device only the leaves of those meshes are re-baked and only their ancestors refitted
 * (and re-quantised), the result being the tree rt_update_vertices would leave, bit for bit.
 * A few moved meshes cost one small launch (0.07 ms for one 720-triangle mesh of a 7.2 M-triangle scene).  The list of
 * tree nodes above a mesh range is kept on the device: a call that names the same range as an earlier one (an animation)
 * sends its vertices and launches, nothing else (50 such meshes: 0.24 ms per call); a long list is made on the first
 * repeat of a range, until then the whole-tree launches serve (0.6 ms).
 * No reference counterpart (the reference has no incremental path at all). */
int rt_update_vertex_range(rt_ctx *ctx, const rt_vertex *vertices, size_t first_vertex, size_t n_vertices,
                           const rt_mesh_info *meshes, size_t first_mesh, size_t n_meshes);

/* ---- view: replaces bind group 0 (View uniform, dynamic offset) and
 * set_camera_viewport (src/node.rs:58-62,71-76; ray_tracing.wgsl:5,176-177).
 * world_from_clip: column-major f32 4x4.  viewport = {x,y,w,h} or NULL for the full target. */
int rt_set_view(rt_ctx *ctx, const float world_from_clip[16], const float world_position[3],
                uint32_t width, uint32_t height, const uint32_t viewport[4]);

/* ---- render: replaces render_pass.draw(0..3, 0..1) (src/node.rs:77) ------------------- */
#define RT_FLAG_SHADOW        0x0001u  /* trace the build-defined shadow ray toward +Y */
#define RT_FLAG_NO_MESH_CULL  0x0002u  /* extension: ignore the per-mesh AABB cull (wgsl:155) */
#define RT_FLAG_BRUTE_FORCE   0x0004u  /* reference-shaped two-level loop (wgsl:150-170), no BVH */
#define RT_FLAG_COUNTERS      0x0008u  /* instrumented kernels: count node visits / tests */
#define RT_FLAG_PATH_TRACE    0x0010u  /* progressive path tracing (wgsl:180-202 loop shape); frames of at most
                                          2^26 pixels (8K is 2^25): RT_ERR_ARG beyond */
#define RT_FLAG_ACCUMULATE    0x0020u  /* keep the accumulation buffer from earlier frames */
#define RT_FLAG_TIME_KERNELS  0x0040u  /* bracket kernels with HIP events (rt_get_counters) */
#define RT_FLAG_STATIC_GRID   0x0100u  /* primary/shadow rays: one thread per pixel on a hardware-scheduled grid
                                          (k_primary) instead of the default persistent waves fed from per-XCD
                                          work queues (k_stream); results are identical (measured slower) */
#define RT_FLAG_WAVE_TRAVERSAL 0x0080u /* primary/shadow rays: wave-coherent (packet) BVH traversal, one
                                          node fetch per wave, instead of per-lane traversal; results
                                          are identical (measured slower on the bench scenes) */
#define RT_FLAG_BINARY_TREE   0x0200u  /* walk the binary tree even when the 4-wide quantised tree was built
                                          (results are identical; for A/B measurements and tests) */

typedef struct {
    uint32_t struct_size;    /* = sizeof(rt_render_params) */
    uint32_t flags;
    uint32_t frame_count;    /* the `count` uniform (src/ray_tracing.rs:86-87); informational */
    uint32_t spp;            /* path tracing: samples this call */
    uint32_t sample_first;   /* global index of this call's first sample (sample slicing) */
    uint32_t max_segments;   /* path tracing: ray segments per path (reference shape: 4) */
    uint32_t seed;
    uint32_t stream_id;      /* RNG stream (= rank for sample slicing) */
    uint32_t shard_rank;     /* screen-tile sharding: this context renders the tiles whose */
    uint32_t shard_count;    /*   index % shard_count == shard_rank; 0 or 1 = all tiles  */
    uint32_t tile_size;      /* tile edge in pixels, a multiple of 16; 0 = 64 when shard_count > 1 (the tiles ranks own),
                                32 otherwise (then a tile is only the unit dealt to the kernel's work queues) */
    uint32_t _reserved;
} rt_render_params;

/* Asynchronous on the context's stream.  params may be NULL (one primary ray per pixel,
 * the live path of ray_tracing.wgsl:203-211). */
int rt_render(rt_ctx *ctx, const rt_render_params *params);
int rt_synchronize(rt_ctx *ctx);

/* ---- the same frame, allocating nothing ------------------------------------------------------------------------
 * No counterpart in the reference (its node creates new GPU buffers every frame, src/node.rs:42-50).  For callers
 * that cannot allocate: code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.
 * rt_render_static renders what rt_render(params) would -- same kernels, same bits -- but
 *   - writes into the CALLER's device buffers: d_radiance (n_pixels * 4 f32), d_prim_ids (n_pixels u32), d_hit_t
 *     (n_pixels f32), all required; d_accum (n_pixels * 4 f32, the running sums) required with RT_FLAG_PATH_TRACE.
 *     n_pixels must be width * height of the current rt_set_view.  Pixels the call does not cover (viewport, other
 *     ranks' tiles) are cleared on every call.  Without RT_FLAG_ACCUMULATE d_accum restarts from zero; the number of
 *     samples it holds is kept in the context.
 *   - keeps the frame's scratch (work-queue heads, counters, path queues, contribution plane) in a static device
 *     array of RT_STATIC_SCRATCH_BYTES = 32 MiB that every GPU the library runs on holds until the process ends.
 *     Primary / shadow frames use 1.2 KB of it at any frame size; path tracing uses 104 B per pixel (one sample per
 *     launch), so frames beyond RT_STATIC_MAX_PATH_PIXELS (322,560; 640x360 fits) return RT_ERR_ARG with a message
 *     that says so.  RT_FLAG_TIME_KERNELS is refused (it creates events).
 *   - calls no hipMalloc / hipFree / hipEventCreate or any other allocating routine, first call included (scene and
 *     view are set up beforehand with rt_upload_scene / rt_set_view, which do allocate).
 * Asynchronous on ONE stream per context, whatever rt_render did in between: the caller's stream (rt_set_stream) or, on the
 * context's own streams, the stream of frame set 0 (with frames in flight the context's "frame rendered last" then names set
 * 0); rt_get_counters then reports this frame.  ALL calls on one GPU share that GPU's array: calls that overlap on one GPU
 * (two contexts, two streams) race -- serialise them.  The array keeps the last call's contents; what must start at zero is
 * cleared by the call itself.
 * Under stream capture (the call recorded into a HIP graph): hand the capturing stream to rt_set_stream BEFORE the capture
 * begins (rt_set_stream synchronises, which a capture forbids) -- a call on any other stream is simply not recorded.  The
 * recorded launches hold BY VALUE: the device addresses of the scene's buffers, the view (camera matrix, size, viewport), the
 * render parameters, the caller's output pointers, the address of the context's pinned counter block and of this GPU's
 * scratch array.  A replay therefore redoes exactly the recorded frame (rt_update_vertices / rt_update_vertex_range between
 * replays are seen: they rewrite the scene's buffers in place) and is valid until the next rt_upload_scene* or rt_destroy of
 * this context, or until the caller's buffers are freed; replaying later is a use after free.  A new camera or new
 * parameters need a new capture.  RT_FLAG_ACCUMULATE is refused under capture (RT_ERR_ARG): the sample count that divides
 * the running sums is host state.  Replays on one GPU share the scratch array like any other calls: one at a time.
 * (tests/test_zz_no_alloc.py captures a primary + shadow frame and a path-traced frame with torch.cuda.graph and replays
 * each three times, bit-identical to rt_render.) */
#define RT_STATIC_SCRATCH_BYTES   (32u << 20)
#define RT_STATIC_MAX_PATH_PIXELS 322560u
int rt_render_static(rt_ctx *ctx, const rt_render_params *params, void *d_radiance, void *d_prim_ids, void *d_hit_t,
                     void *d_accum, size_t n_pixels);

/* ---- results: replace the colour-attachment write (src/node.rs:63-69) -----------------
 * Caller-allocated host buffers; each call synchronises the stream first.
 * radiance: W*H*4 f32 linear unclamped RGBA (path tracing: accumulated sum / samples).
 * prim ids: W*H u32, global triangle index of the primary hit, 0xFFFFFFFF on a miss.
 * hit t:    W*H f32, 0 on a miss.
 * bgra8:    W*H*4 u8, clamp + sRGB encode as the Bgra8UnormSrgb target (src/pipeline.rs:56-60). */
int rt_read_radiance(rt_ctx *ctx, float *rgba, size_t n_floats);
int rt_read_prim_ids(rt_ctx *ctx, uint32_t *prim, size_t n);
int rt_read_hit_t(rt_ctx *ctx, float *t, size_t n);
int rt_read_bgra8_srgb(rt_ctx *ctx, uint8_t *bgra, size_t n_bytes);
int rt_read_accumulation(rt_ctx *ctx, float *accum, size_t n_floats);
/* The same Bgra8UnormSrgb encode (src/pipeline.rs:56-60) for ANY linear RGBA f32 frame the caller holds on the host
 * (e.g. one combined from several GPUs): n_pixels * 4 floats in, n_pixels * 4 bytes out, through the device kernel.
 * Byte work, exact: a colour byte is the correctly rounded round(255 * srgb_oetf(x)) (threshold table generated by
 * tools/gen_srgb_table.py), alpha is floor(255 * clamp(a) + 0.5); NaN encodes as 0. */
int rt_encode_bgra8_srgb(rt_ctx *ctx, const float *rgba, size_t n_pixels, uint8_t *bgra);
/* Presentation that does not stall frames in flight (what the colour-attachment write of src/node.rs:63-69 is to a
 * render graph that keeps submitting).  rt_present_async encodes the frame rendered LAST into Bgra8UnormSrgb bytes and
 * enqueues their copy into host_bgra (W*H*4 bytes; pinned or registered host memory for a copy that really is
 * asynchronous), all on that frame's stream, and returns at once; *ticket names the frame's set (0..3).
 * rt_present_wait blocks until the bytes of that ticket have arrived.  A ticket is good until its set is presented
 * again, i.e. for the next n - 1 calls with n frames in flight; host_bgra must stay untouched until its wait returns.
 * The bytes are those rt_read_bgra8_srgb returns for the same frame. */
int rt_present_async(rt_ctx *ctx, uint8_t *host_bgra, size_t n_bytes, uint32_t *ticket);
int rt_present_wait(rt_ctx *ctx, uint32_t ticket);

/* Device-side access for callers that keep results on the GPU (multi-GPU combine with RCCL):
 * pointers into the context's own buffers, valid until the next rt_set_view with a different
 * size or rt_destroy. */
int rt_device_buffers(rt_ctx *ctx, void **d_radiance, void **d_prim, void **d_accum);
/* radiance = accum / total_samples on the device (after an external reduce). */
int rt_resolve_accumulation(rt_ctx *ctx, uint32_t total_samples);

/* ---- multi-GPU tile exchange (the reference is single-GPU, src/node.rs:77 is its one launch) --------
 * With shard_count ranks, rank r owns the tile_size x tile_size screen tiles g = r, r + shard_count, ... (g counted
 * row-major over the covered rect), exactly the tiles rt_render(shard_rank = r) writes.  rt_pack_tiles copies the
 * radiance of this context's own tiles into ONE contiguous device buffer (tile after tile in owned order, tile_size^2
 * float4 each, zeros where an edge tile leaves the covered rect) -- the message a rank sends to the gathering rank;
 * rt_unpack_tiles scatters the packed tiles of n_ranks ranks, slot_bytes apart in d_packed (slot s = rank
 * first_rank + s), into this context's radiance buffer.  rt_tile_slot_bytes = the packed size of the rank that owns
 * the most tiles (rank 0): a slot size that fits every rank.  Device pointers; asynchronous on the context's stream. */
int rt_tile_slot_bytes(rt_ctx *ctx, uint32_t shard_count, uint32_t tile_size, size_t *bytes);
int rt_pack_tiles(rt_ctx *ctx, uint32_t shard_rank, uint32_t shard_count, uint32_t tile_size, void *d_packed, size_t bytes);
int rt_unpack_tiles(rt_ctx *ctx, uint32_t first_rank, uint32_t n_ranks, uint32_t shard_count, uint32_t tile_size,
                    const void *d_packed, size_t slot_bytes);

/* ---- counters, introspection, errors ------------------------------------------------- */
typedef struct {
    uint64_t rays_primary, rays_shadow, rays_bounce;
    uint64_t node_visits;        /* BVH nodes fetched (RT_FLAG_COUNTERS only) */
    uint64_t tri_tests;          /* ray/triangle tests (RT_FLAG_COUNTERS only) */
    uint64_t mesh_aabb_tests;    /* per-mesh culls evaluated (RT_FLAG_COUNTERS only) */
    double   kernel_ms;          /* RT_FLAG_TIME_KERNELS: sum of the begin-to-end spans (HIP event pairs) of the timed frames
                                    since the last rt_get_counters.  One frame at a time that is device time; with frames in
                                    flight the spans of different sets OVERLAP, so the sum exceeds wall time by up to n */
    uint32_t kernel_launches;
    uint32_t _pad;
    uint64_t wave_node_iters;    /* wave-level trips of the node loop / of the triangle loop */
    uint64_t wave_tri_iters;     /*   (RT_FLAG_COUNTERS, per-lane kernels): lane utilisation */
    uint64_t wave_alive_lanes;   /* sum over node-loop trips of the wave's unfinished rays */
} rt_counters;
int rt_get_counters(rt_ctx *ctx, rt_counters *out);

typedef struct {
    uint64_t n_nodes, n_triangles, n_meshes, n_vertices;   /* n_triangles: leaf records (triangles + spheres) */
    uint32_t node_bytes, tri_bytes;  /* bytes per BVH node / per leaf triangle record */
    uint32_t max_depth, max_leaf;
    double   build_ms;               /* of the last build or refit */
    double   sah_cost;               /* of the last full build */
    uint32_t n_builds, n_refits;     /* since rt_create */
    uint64_t n_wide_nodes;           /* 4-wide quantised tree over the same records (0: not built); 64 B per node */
    uint32_t wide_max_depth, wide_stack_need;
} rt_bvh_info;
int rt_get_bvh_info(rt_ctx *ctx, rt_bvh_info *out);
/* Copy the flattened BVH out (host buffers sized from rt_get_bvh_info). */
int rt_read_bvh(rt_ctx *ctx, void *nodes, size_t node_bytes_total, void *tris, size_t tri_bytes_total);
/* Copy the 4-wide tree out: n_wide_nodes * 64 bytes (node layout: ray-tracing_amd/csrc/rt_layout.h RtWideNode). */
int rt_read_bvh_wide(rt_ctx *ctx, void *wide_nodes, size_t bytes_total);

/* Host-only BVH build over the same arrays (no GPU touched): lets the builder and the flattened
 * layout be checked on a machine without a device.  Sizes from *info, then copy, then free. */
typedef struct rt_host_bvh rt_host_bvh;
int rt_host_bvh_build(const rt_triangle *triangles, size_t n_triangles, const rt_mesh_info *meshes, size_t n_meshes,
                      const rt_vertex *vertices, size_t n_vertices, int n_threads, rt_host_bvh **out, rt_bvh_info *info);
int rt_host_bvh_refit(rt_host_bvh *bvh, const rt_triangle *triangles, size_t n_triangles,
                      const rt_vertex *vertices, size_t n_vertices, int n_threads);
int rt_host_bvh_copy(const rt_host_bvh *bvh, void *nodes, size_t node_bytes_total, void *tris, size_t tri_bytes_total);
/* The 4-wide quantised tree made from the (possibly refitted) binary tree: 64-B nodes and, per node, the four
 * source slots (16 B).  Call with NULL buffers to get the node count, then with buffers of that many entries.
 * stack_need: most entries a traversal of it can hold on its stack at once. */
int rt_host_bvh_wide(const rt_host_bvh *bvh, void *wide_nodes, void *wide_src, size_t capacity_nodes,
                     size_t *n_wide, uint32_t *stack_need);
void rt_host_bvh_free(rt_host_bvh *bvh);

/* ---- measurement aid (no counterpart in the reference; not on the render path) -------------------------------------------
 * What THIS GPU's memory system delivers to the access pattern of a BVH walk -- every lane one 64-B record (4 x dwordx4) at
 * its own pseudo-random address -- per level of the hierarchy and at the traversal kernels' occupancy: the denominators
 * bench.py's roofline.gather uses instead of the HBM streaming peak.  Fills ggather_per_s[0..9] with 10^9 gathers of 64 B per
 * second, as pairs {independent (throughput), dependent (pointer chase)} for buffers of 16 KiB (vector L1), 2 MiB (L2 of every
 * XCD), 128 MiB (Infinity Cache), 1 GiB and 4 GiB (HBM).  Allocates 4 GiB of device memory for the call (RT_ERR_OOM if it
 * cannot) and takes about a second; runs in the caller's process on a stream of its own and leaves the thread's current
 * device as it found it. */
#define RT_GATHER_RATE_VALUES 10
int rt_measure_gather_rates(int device_ordinal, double *ggather_per_s, size_t n_values);

const char *rt_last_error(rt_ctx *ctx);   /* ctx may be NULL: last error of rt_create */
const char *rt_status_string(int status);
const char *rt_version(void);

#ifdef __cplusplus
}
#endif
#endif /* RT_AMD_H */
