// bvh_build.h -- host BVH builder (binned SAH, multithreaded).  Build-side design: the
// reference's src/bvh.rs:1 is empty, its only acceleration is one AABB per mesh
// (src/ray_tracing.rs:148-156).
#pragma once
#include <vector>
#include "rt_layout.h"

struct RtBvhBuildResult {
    std::vector<RtBvhNode> nodes;     // nodes[0] is the root when non-empty
    std::vector<RtTriRecord> recs;    // leaf order
    uint32_t max_depth = 0, max_leaf = 0;
    double sah_cost = 0.0, build_ms = 0.0;
    uint64_t dropped = 0;             // records dropped for non-finite vertices
};

// Flattens the (mesh, triangle) loop of ray_tracing.wgsl:153,158 into records and builds the
// tree.  Inputs must already be validated (indices in range).  n_threads <= 0: hardware
// concurrency capped at 16.
// Spheres (extension) become records after the triangles' (seq continues), boxed centre +/- radius.
void rt_bvh_build(const rt_triangle *tris, size_t n_tris, const rt_mesh_info *meshes, size_t n_meshes,
                  const rt_vertex *verts, size_t n_verts, int n_threads, RtBvhBuildResult *out,
                  const rt_sphere *spheres = nullptr, size_t n_spheres = 0);

// f1: vertices moved, topology unchanged.  Re-bakes every leaf record from the new vertex
// positions and refits all boxes bottom-up in place (children are stored after their parent, so one
// reverse sweep suffices).  Returns false when a vertex became non-finite (the tree would have to drop
// that triangle): the caller then rebuilds.
bool rt_bvh_refit(RtBvhBuildResult *bvh, const rt_triangle *tris, const rt_vertex *verts, int n_threads,
                  const rt_sphere *spheres = nullptr, size_t n_tris = 0);

// 4-wide quantised tree over the same records (rt_layout.h RtWideNode): the binary tree collapsed greedily --
// starting from a node's two children, the inner child with the largest box is replaced by its own two children
// until four slots are used -- and the child boxes quantised (rt_wide.h).  src keeps, per slot, which binary-tree
// box it was made from, so that a refit of the binary tree is followed by a plain re-quantisation.
struct RtWideTree {
    std::vector<RtWideNode> nodes;    // nodes[0] is the root; children are stored after their parent
    std::vector<RtWideSrc> src;
    uint32_t max_depth = 0;
    uint32_t stack_need = 0;          // most entries a traversal can have on its stack at once
    double build_ms = 0.0;
};
void rt_bvh_widen(const RtBvhBuildResult &bvh, RtWideTree *out);
void rt_wide_requantise(const RtBvhBuildResult &bvh, RtWideTree *w);   // after rt_bvh_refit
