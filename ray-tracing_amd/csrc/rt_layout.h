// rt_layout.h -- device-side data layout shared by the host runtime, the BVH builder and the
// HIP kernels.  Build-side design: the reference's src/bvh.rs is an empty file (SURVEY.md
// section 0), so the acceleration structure and its node layout originate here.
#pragma once
#include <stdint.h>
#include "../../include/rt_amd.h"

// One BVH node = 64 B = half a 128-B L2 line, fetched by a lane as 4 x dwordx4.
// A node stores the boxes of BOTH children, so one fetch decides both descents.
//   q0 = lo0.xyz, hi0.x      q1 = hi0.yz, lo1.xy     q2 = lo1.z, hi1.xyz
//   q3 = child0, child1 (int bits), 2 spare dwords
// child >= 0 : index of an inner node
// child <  0 : leaf; code = ~child = (first_record << 3) | (count - 1), count in 1..8
// An absent child has a NaN box (never intersected) and child = RT_CHILD_NONE.
struct alignas(16) RtBvhNode {
    float lo0[3], hi0[3];
    float lo1[3], hi1[3];
    int32_t child0, child1;
    uint32_t pad0, pad1;
};
static_assert(sizeof(RtBvhNode) == 64, "node must be 64 B");
#define RT_CHILD_NONE ((int32_t)0x80000000)
#define RT_MAX_LEAF 4
#define RT_MAX_DEPTH 48   // builder guarantee; the kernels size their stacks from it
#define RT_STACK_CAP 80   // entries of the per-lane traversal stack (LDS part + private part)

// Wide node: up to FOUR children, their boxes as 8-bit offsets on a per-node grid -- 64 B as well, so one fetch
// decides four descents and the dependent fetches per ray roughly halve.  Plane of child k on axis a:
//   org[a] + byte_k(lo_a or hi_a) * scl[a]          (byte k of the packed word = child k)
// The grid boxes CONTAIN the binary tree's boxes they were made from (rt_wide.h: expanded by 2e-6 of the
// coordinate magnitude, then floor / ceil), so culling stays conservative and results do not change.
// child[k] uses the binary tree's codes: >= 0 index of a WIDE node, < 0 leaf code, RT_CHILD_NONE = empty slot.
//   q0 = org.xyz, scl.x    q1 = scl.yz, lo_x, lo_y    q2 = lo_z, hi_x, hi_y, hi_z    q3 = child[0..3]
struct alignas(16) RtWideNode {
    float org[3]; float sclx;
    float scly, sclz; uint32_t lox, loy;
    uint32_t loz, hix, hiy, hiz;
    int32_t child[4];
};
static_assert(sizeof(RtWideNode) == 64, "wide node must be 64 B");
// Where the four boxes of a wide node come from: (binary node index << 1) | child side; RT_WIDE_EMPTY = none.
struct RtWideSrc { uint32_t slot[4]; };
#define RT_WIDE_EMPTY 0xFFFFFFFFu
// The wide step computes byte*(scl/d) - (o - org)/d with 1/d clamped to 1e30: every term stays finite only while the
// coordinates do.  Scenes or camera positions beyond this magnitude are walked on the binary tree, whose
// (plane - o)/d form has no such limit.
#define RT_WIDE_COORD_MAX 5e7f

// One leaf triangle record = 48 B, stored in BVH leaf order so a leaf is one contiguous run.
// a, ab = b-a, ac = c-a are the exact f32 values ray_triangle computes first
// (ray_tracing.wgsl:119-123), precomputed once instead of 3 dependent vertex gathers per test.
//   tri  : index into the caller's Triangle array (T1)  -> reported primitive id
//   mesh : index into the caller's MeshInfo array (T2)  -> material + per-mesh AABB cull
//   seq  : position in the reference's loop order (mesh order, then triangle order,
//          ray_tracing.wgsl:153,158); equal-t ties resolve to the lowest seq (wgsl:164)
struct alignas(16) RtTriRecord {
    float a[3];  float abx;
    float aby, abz, acx, acy;
    float acz; uint32_t tri, mesh, seq;
};
static_assert(sizeof(RtTriRecord) == 48, "tri record must be 48 B");
// An analytic sphere (extension) uses the same 48-B slot: a = centre, abx = radius, the other edge
// floats 0, tri = n_triangles + sphere index (its primitive id), mesh = RT_REC_SPHERE, seq after
// every triangle's.
#define RT_REC_SPHERE 0xFFFFFFFFu

static_assert(sizeof(rt_triangle) == 12, "T1");
static_assert(sizeof(rt_mesh_info) == 48, "T2");
static_assert(sizeof(rt_vertex) == 32, "T3");
static_assert(sizeof(rt_material) == 16, "T4");

struct RtSceneDev {
    const RtBvhNode *nodes;
    const RtTriRecord *recs;
    const rt_triangle *tris;
    const rt_mesh_info *meshes;
    const rt_vertex *verts;
    const rt_material *mats;
    const rt_sphere *spheres;          // extension: analytic spheres and their material indices
    const uint32_t *sphere_mats;
    const RtWideNode *wide;            // 4-wide quantised tree over the same records (n_wide = 0: not built)
    uint32_t n_nodes, n_recs, n_tris, n_meshes, n_verts, n_mats, n_spheres, n_wide;
};

struct RtViewDev {
    float m[16];          // world_from_clip, column-major
    float pos[3];         // world_position
    uint32_t width, height;
    uint32_t vx, vy, vw, vh;  // covered rect (the viewport, or the whole target)
};

struct RtShardDev {
    uint32_t rank, count, tile;   // tile edge in px; count <= 1 => no sharding
    uint32_t tiles_x, tiles_y;    // shard tiles over the covered rect
    uint32_t n_owned;             // shard tiles owned by this rank
};

// device counters block (64-bit each); index constants
enum { RT_CNT_SHADOW = 0, RT_CNT_BOUNCE = 1, RT_CNT_NODES = 2, RT_CNT_TRIS = 3, RT_CNT_MESH = 4,
       RT_CNT_QUEUE_A = 5, RT_CNT_QUEUE_B = 6, RT_CNT_PRIMARY = 7, RT_CNT_WORK = 8, RT_CNT_WNODE = 9, RT_CNT_WTRI = 10, RT_CNT_WALIVE = 11,
       RT_CNT_HEADS = 16,            // 8 per-XCD work-queue heads, one 128-B line each (16 x u64 apart)
       RT_CNT_N = 16 + 8 * 16 };

// path-tracing wavefront queue, SoA: 11 dwords per path
struct RtPathQueue {
    float *ox, *oy, *oz, *dx, *dy, *dz, *cr, *cg, *cb;
    uint32_t *rng, *pixel;
};

