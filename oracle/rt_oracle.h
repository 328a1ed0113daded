/*
 * rt_oracle.h -- CPU ORACLE for the per-pixel ray-tracing hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, the smoke check in
 * __graft_entry__.py and the cpu_baseline leg of bench.py may load it.  The product
 * library (ray-tracing_amd/csrc) never includes, links or calls anything in oracle/.
 *
 * PARITY UNPINNED BY THE REFERENCE: the reference (camms205/ray-tracing) ships no tests,
 * golden vectors or fixtures and cannot be compiled or run here (no Rust toolchain, no
 * WGSL runtime).  This file is a literal scalar restatement of
 *   assets/shaders/ray_tracing.wgsl:118-139  (ray_triangle)
 *   assets/shaders/ray_tracing.wgsl:141-148  (ray_aabb)
 *   assets/shaders/ray_tracing.wgsl:150-170  (hit_triangles)
 *   assets/shaders/ray_tracing.wgsl:173-177,203-211 (ray generation, shade)
 *   assets/shaders/ray_tracing.wgsl:38-74    (RNG)
 *   assets/shaders/ray_tracing.wgsl:180-202  (commented path-trace loop shape)
 *   src/ray_tracing.rs:98-169                (prepare_meshinfo)
 * plus a scalar restatement of the product's tree walk (orc_hit_tree, below; no reference counterpart) that is used to
 * cross-check the GPU's counted node visits / triangle tests and as a second, tree-based CPU figure;
 * pinned by the hand-worked known-answer tests KAT-1..6 of SURVEY.md section 4
 * (tests/test_oracle_kat.py) and by golden images it produced itself
 * (tests/golden/, made by tests/golden/make_golden.py).
 *
 * Float rules (fixed here, mirrored by the HIP kernels; DESIGN.md "Float contract"):
 *   - IEEE binary32, round-to-nearest-even, no FMA contraction (-ffp-contract=off),
 *     subnormals kept, division and sqrt correctly rounded.
 *   - dot(a,b)   = (a.x*b.x + a.y*b.y) + a.z*b.z
 *   - cross(a,b) = (a.y*b.z - a.z*b.y, a.z*b.x - a.x*b.z, a.x*b.y - a.y*b.x)
 *   - normalize(v) = v / sqrt(dot(v,v)), component-wise division
 *   - min/max are IEEE fminf/fmaxf (a NaN operand is dropped)
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* T1  src/ray_tracing.rs:51-54, wgsl:19-23 -- 12 B */
typedef struct { uint32_t idx[3]; } orc_triangle;
/* T2  src/ray_tracing.rs:56-63, wgsl:24-30 -- 48 B (std430: vec3 aligned to 16) */
typedef struct {
    uint32_t first_tri, tri_count, material, pad0;
    float aabb_min[3]; float pad1;
    float aabb_max[3]; float pad2;
} orc_mesh_info;
/* T3  src/ray_tracing.rs:92-93, wgsl:14-17 -- 32 B */
typedef struct { float pos[3]; float pad0; float norm[3]; float pad1; } orc_vertex;
/* T4  src/ray_tracing.rs:65-68, wgsl:32-34 -- 16 B */
typedef struct { float color[4]; } orc_material;

/* Analytic sphere (EXTENSION, SURVEY 8d config 2 variant ii): centre + radius as the dead
 * Sphere type of src/scene.rs:26-28 has them; the reference has no ray/sphere test (its live
 * spheres are tessellated, src/main.rs:85), so this part of the oracle is BUILD-DEFINED and
 * parity for it is unpinned by the reference.  16 B. */
typedef struct { float center[3]; float radius; } orc_sphere;

typedef struct {
    const orc_triangle *tris;    size_t n_tri;
    const orc_mesh_info *meshes; size_t n_mesh;
    const orc_vertex *verts;     size_t n_vtx;
    const orc_material *mats;    size_t n_mat;
    /* extension: analytic spheres, tested after all meshes (prim id = n_tri + index) */
    const orc_sphere *spheres;   const uint32_t *sphere_mats; size_t n_sphere;
} orc_scene;

/* Subset of Bevy's View uniform the shader reads (wgsl:176-177). Column-major mat4. */
typedef struct {
    float world_from_clip[16];
    float world_position[3];
    uint32_t width, height;          /* full target size */
    uint32_t vp_x, vp_y, vp_w, vp_h; /* viewport rect (src/node.rs:71-73); vp_w==0 -> full */
} orc_view;

typedef struct {
    int hit;
    float point[3];
    float normal[3];
    float t;
    uint32_t material;
    uint32_t prim;      /* global triangle index (build-side addition for parity checks) */
} orc_hit;

#define ORC_FLAG_SHADOW        1u  /* build-defined shadow ray toward +Y */
#define ORC_FLAG_NO_MESH_CULL  2u  /* extension: skip the per-mesh AABB cull (wgsl:155) */
#define ORC_MISS_PRIM 0xFFFFFFFFu
#define ORC_SHADOW_ATTEN 0.25f
#define ORC_SHADOW_EPS   1e-3f

/* --- per-ray pieces -------------------------------------------------------------- */
void orc_ray_gen(const orc_view *v, uint32_t px, uint32_t py, float o[3], float d[3]);
int  orc_ray_aabb(const float o[3], const float d[3], const float lb[3], const float rt[3]);
void orc_ray_triangle(const orc_scene *s, const float o[3], const float d[3], uint32_t tri, orc_hit *out);
void orc_hit_triangles(const orc_scene *s, const float o[3], const float d[3], uint32_t flags, orc_hit *out);
/* Build-defined ray/sphere test (no reference counterpart): oc = o - c; a = dot(d,d); hb = dot(oc,d);
 * cc = dot(oc,oc) - r*r; disc = hb*hb - a*cc; t = (-hb - sqrt(disc)) / a; hit iff disc >= 0 && t >= 0
 * (entry point only: an origin inside the sphere misses, as back faces do for triangles, wgsl:137);
 * normal = normalize(point - c). */
void orc_ray_sphere(const orc_scene *s, const float o[3], const float d[3], uint32_t sphere, orc_hit *out);
void orc_shade(const orc_scene *s, const orc_hit *h, float rgba[4]);

/* --- whole frames ---------------------------------------------------------------- */
/* Live path (wgsl:173-211): one primary ray per pixel (+ optional shadow ray).
 * rgba: W*H*4 f32 linear, unclamped; prim: W*H u32; t: W*H f32 (0 on miss). Any may be NULL.
 * If pixel_list != NULL only those n_pixels linear pixel indices are traced and the outputs
 * are indexed by list position (used for the bounded cpu_baseline sample).
 * Returns the number of rays traced (primary + shadow). */
uint64_t orc_render_primary(const orc_scene *s, const orc_view *v, uint32_t flags,
                            float *rgba, uint32_t *prim, float *t,
                            const uint32_t *pixel_list, size_t n_pixels, int n_threads);

/* --- tree walk ("cross-checked by the CPU BVH oracle, which traverses in the same order") ------
 * NO REFERENCE COUNTERPART (src/bvh.rs:1 is empty): a scalar, one-ray-at-a-time restatement of the walk the
 * product's kernels make over the product's acceleration tree -- same child order, same slab arithmetic, same
 * closest-hit rule (lowest t, then lowest position in the reference's loop order, wgsl:164), same per-mesh cull
 * (wgsl:155) -- so that (1) the walk can be checked against orc_hit_triangles hit for hit on the CPU, and (2) the
 * node visits / triangle tests the GPU counts for the roofline (rt_get_counters) can be checked to the unit.
 * The tree is DATA handed in by the test (built by the product's host builder, read through rt_host_bvh /
 * rt_read_bvh); the layouts are restated here, not included:
 *   binary node, 64 B: lo0[3] hi0[3] lo1[3] hi1[3] (f32), child0, child1 (i32), 2 spare words
 *   wide node,   64 B: org[3] scl[3] (f32), lo_x lo_y lo_z hi_x hi_y hi_z (u32, byte k = child k), child[4] (i32)
 *   leaf record, 48 B: a[3] ab[3] ac[3] (f32), tri, mesh, seq (u32); mesh = 0xFFFFFFFF: sphere (centre, radius in a, ab.x)
 *   child code: >= 0 node index, < 0 leaf ~((first_record << 3) | (count - 1)), 0x80000000 absent                 */
typedef struct {
    const void *nodes;  size_t n_nodes;   /* binary nodes */
    const void *wide;   size_t n_wide;    /* wide nodes; n_wide != 0 => the wide tree is the one walked */
    const void *recs;   size_t n_recs;
} orc_tree;
typedef struct { uint64_t node_visits, tri_tests, mesh_culls; } orc_walk_counts;
/* closest hit (any_hit = 0) or first accepted hit (any_hit = 1, shadow rays); counts are ADDED to *cnt */
void orc_hit_tree(const orc_scene *s, const orc_tree *tr, const float o[3], const float d[3], uint32_t flags,
                  int any_hit, orc_hit *out, orc_walk_counts *cnt);
/* orc_render_primary with orc_hit_tree in place of orc_hit_triangles; *cnt (may be NULL) receives the sums */
uint64_t orc_render_primary_tree(const orc_scene *s, const orc_tree *tr, const orc_view *v, uint32_t flags,
                                 float *rgba, uint32_t *prim, float *t,
                                 const uint32_t *pixel_list, size_t n_pixels, int n_threads, orc_walk_counts *cnt);

/* --- RNG (wgsl:38-74), integer core + FP wrappers ---------------------------------- */
uint32_t orc_next_random(uint32_t *state);
float orc_rand(uint32_t *state);
float orc_det_logf(float x);   /* deterministic log / cos built from IEEE basic ops only */
float orc_det_cosf(float x);
float orc_rand_norm(uint32_t *state);
void  orc_rand_dir(uint32_t *state, float d[3]);
void  orc_rand_hemi(uint32_t *state, const float n[3], float d[3]);
uint32_t orc_seed(uint32_t pixel, uint32_t sample, uint32_t seed, uint32_t stream);

/* Progressive path tracing, loop shape of wgsl:180-202 (dead code in the reference;
 * build-defined where the reference leaves holes -- DESIGN.md "Path tracing contract").
 * accum: W*H*4 f32 running sums, updated in place sample by sample (alpha accumulates 1 per
 * sample). Returns the number of ray segments traced. */
uint64_t orc_render_path(const orc_scene *s, const orc_view *v, uint32_t flags,
                         uint32_t spp, uint32_t sample_first, uint32_t max_segments,
                         uint32_t seed, uint32_t stream, float *accum, int n_threads);
/* the same through the tree walk (orc_hit_tree in place of orc_hit_triangles): scenes of a size the
 * reference-shaped loop cannot reach; *cnt (may be NULL) receives the walk's sums */
uint64_t orc_render_path_tree(const orc_scene *s, const orc_tree *tr, const orc_view *v, uint32_t flags,
                              uint32_t spp, uint32_t sample_first, uint32_t max_segments,
                              uint32_t seed, uint32_t stream, float *accum, int n_threads, orc_walk_counts *cnt);
/* the same for the pixels of one rectangle of the target only (crop = x, y, w, h; NULL: all): every pixel gets the
 * value it has in the whole frame (seeds and rays depend on the pixel, not on the crop), so a few crops check a frame
 * whose whole would take the host minutes.  Test infrastructure of the full-size configs[3] / configs[4] cases. */
uint64_t orc_render_path_tree_crop(const orc_scene *s, const orc_tree *tr, const orc_view *v, uint32_t flags,
                                   uint32_t spp, uint32_t sample_first, uint32_t max_segments,
                                   uint32_t seed, uint32_t stream, const uint32_t crop[4], float *accum, int n_threads,
                                   orc_walk_counts *cnt);
/* out = accum / total_samples (alpha too). */
void orc_resolve(const float *accum, size_t n_px, uint32_t total_samples, float *rgba);

/* --- presentation: linear f32 -> clamp -> sRGB -> BGRA8 (src/pipeline.rs:56-60) ----- */
void orc_encode_bgra8_srgb(const float *rgba, size_t n_px, uint8_t *bgra);

/* --- scene extraction (src/ray_tracing.rs:98-169) --------------------------------- */
typedef struct {
    const float *positions;   /* n_vtx*3, NULL => "missing attribute" */
    const float *normals;     /* n_vtx*3, NULL => "missing attribute" */
    uint32_t n_vtx;
    const uint32_t *indices;  /* n_idx */
    uint32_t n_idx;
    float base_color_linear[4];
    float scale[3];
    float rotation[4];        /* quaternion x,y,z,w */
    float translation[3];
    float aabb_half_extents[3];
} orc_entity;

/* Flattens entities exactly as prepare_meshinfo does, quirks kept.
 * Output arrays must be large enough (sum of counts). Counts returned through n_*. */
void orc_prepare_meshinfo(const orc_entity *ents, size_t n_ent,
                          orc_triangle *tris, size_t *n_tri,
                          orc_mesh_info *meshes, size_t *n_mesh,
                          orc_vertex *verts, size_t *n_vtx,
                          orc_material *mats, size_t *n_mat);

#ifdef __cplusplus
}
#endif
#endif
