// bvh_build.cpp -- binned-SAH BVH2 builder over the reference's (mesh, triangle) loop items.
// Build-side design (reference src/bvh.rs:1 is empty).  Host code, no GPU calls.
#include "bvh_build.h"
#include "rt_wide.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <limits>
#include <memory>
#include <thread>

namespace {

// 32 B: the passes over the primitives of the big nodes are memory-bound.  The centroid is 0.5*lo + 0.5*hi wherever it is
// needed; the mesh of a triangle is found again from seq (its position in the reference's loop order) at the end.
struct Prim {
    float lo[3], hi[3];
    uint32_t seq, tri;
};
inline float cen(const Prim &P, int k) { return 0.5f * P.lo[k] + 0.5f * P.hi[k]; }
constexpr uint32_t kDroppedTri = 0xFFFFFFFFu;

struct TmpNode {
    float lo[3], hi[3];
    int32_t left, right;       // inner: indices of TmpNodes; leaf: -1
    uint32_t first, count;     // leaf: run in the prim array
    uint32_t inner;            // inner nodes in this subtree, this one included (0 for a leaf)
};

constexpr int kMaxBins = 64;
int kBins = 16;           // RT_TUNE_BINS (experiments)
// SAH cost model; env overrides are experiment hooks (RT_TUNE_COST_TRI, RT_TUNE_MAX_LEAF)
// kCostTri 2: with four children decided per node fetch a triangle test costs about two node steps; the build then
// splits down to (mostly) single-triangle leaves: bench scene 0.87 -> 0.83 ms, 1 M-triangle soup 0.66 -> 0.64 (1 was
// best for the binary tree)
float kCostNode = 1.0f, kCostTri = 2.0f;
uint32_t kMaxLeaf = RT_MAX_LEAF;
constexpr size_t kSpawnMin = 1u << 15;
constexpr size_t kParallelMin = 1u << 18;   // nodes this big run their passes over the primitives on the spare threads

inline float half_area(const float *lo, const float *hi) {
    float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    return dx * dy + dy * dz + dz * dx;
}
inline void box_reset(float *lo, float *hi) {
    for (int k = 0; k < 3; ++k) { lo[k] = std::numeric_limits<float>::infinity(); hi[k] = -lo[k]; }
}
inline void box_grow(float *lo, float *hi, const float *blo, const float *bhi) {
    for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], blo[k]); hi[k] = std::max(hi[k], bhi[k]); }
}

struct Builder {
    std::vector<Prim> prims;
    std::unique_ptr<TmpNode[]> nodes;      // 2n+1 entries, not zeroed: every node is written by build()
    std::atomic<uint32_t> next_node{0};
    std::atomic<int> free_threads{0};
    std::atomic<uint32_t> max_depth{0};

    uint32_t alloc() { return next_node.fetch_add(1); }

    void note_depth(uint32_t d) {
        uint32_t cur = max_depth.load();
        while (d > cur && !max_depth.compare_exchange_weak(cur, d)) {}
    }

    // spare threads: taken for the passes over a big node's primitives, or for a whole subtree
    int grab(int want) {
        int got = 0;
        while (got < want) {
            if (free_threads.fetch_sub(1) > 0) ++got;
            else { free_threads.fetch_add(1); break; }
        }
        return got;
    }
    template <class F>
    static void split_run(size_t b, size_t e, int parts, F fn) {      // fn(part, begin, end) on `parts` threads
        if (parts <= 1) { fn(0, b, e); return; }
        std::vector<std::thread> th;
        for (int t = 1; t < parts; ++t) th.emplace_back(fn, t, b + (e - b) * (size_t)t / parts, b + (e - b) * (size_t)(t + 1) / parts);
        fn(0, b, b + (e - b) / parts);
        for (auto &t : th) t.join();
    }
    struct Bins {
        uint32_t cnt[3][kMaxBins];
        float blo[3][kMaxBins][3], bhi[3][kMaxBins][3];
        void reset(int bins) {
            for (int ax = 0; ax < 3; ++ax)
                for (int k = 0; k < bins; ++k) { cnt[ax][k] = 0; box_reset(blo[ax][k], bhi[ax][k]); }
        }
    };
    struct Bounds { float lo[3], hi[3], clo[3], chi[3]; };

    // Builds the subtree for prims[b,e) into nodes[ni]; depth = depth of this node (root 0).
    // The passes over the primitives of a big node (bounds, binning) run on every spare thread: minima, maxima and
    // counts do not depend on how the run is cut up, so the tree is the one a single thread builds.
    void build(uint32_t ni, size_t b, size_t e, uint32_t depth) {
        TmpNode &N = nodes[ni];
        size_t n = e - b;
        const int extra = n >= kParallelMin ? grab((int)std::min<size_t>(n / (kParallelMin / 4), 15)) : 0;
        const int parts = 1 + extra;
        float clo[3], chi[3];
        {
            Bounds one; std::vector<Bounds> many;       // no heap traffic for the millions of small nodes
            if (parts > 1) many.resize((size_t)parts);
            Bounds *part = parts > 1 ? many.data() : &one;
            split_run(b, e, parts, [&](int t, size_t pb, size_t pe) {
                Bounds &B = part[(size_t)t];
                box_reset(B.lo, B.hi); box_reset(B.clo, B.chi);
                for (size_t i = pb; i < pe; ++i) {
                    box_grow(B.lo, B.hi, prims[i].lo, prims[i].hi);
                    const float c[3] = {cen(prims[i], 0), cen(prims[i], 1), cen(prims[i], 2)};
                    box_grow(B.clo, B.chi, c, c);
                }
            });
            box_reset(N.lo, N.hi); box_reset(clo, chi);
            for (int t = 0; t < parts; ++t) { box_grow(N.lo, N.hi, part[t].lo, part[t].hi); box_grow(clo, chi, part[t].clo, part[t].chi); }
        }
        note_depth(depth);
        auto make_leaf = [&]() { N.left = N.right = -1; N.first = (uint32_t)b; N.count = (uint32_t)n; N.inner = 0; };
        if (n <= 1) { make_leaf(); return; }

        // depth guard: the inner nodes below this one may use at most `room` more levels
        uint32_t room = RT_MAX_DEPTH - 1 - depth;   // levels available for children inner nodes
        size_t leaves_needed = (n + kMaxLeaf - 1) / kMaxLeaf;
        uint32_t need = 0;
        while (((size_t)1 << need) < leaves_needed) ++need;
        bool force_median = need + 1 >= room;           // a balanced split still fits; SAH might not

        int best_axis = -1, best_bin = -1;
        float best_cost = std::numeric_limits<float>::infinity();
        float parent_area = half_area(N.lo, N.hi);
        if (!force_median) {
            // one pass over the primitives bins all three axes
            bool use[3]; float scale3[3];
            for (int ax = 0; ax < 3; ++ax) {
                float ext = chi[ax] - clo[ax];
                use[ax] = ext > 0.0f;
                scale3[ax] = use[ax] ? (float)kBins / ext : 0.0f;
            }
            // a thread's own bins stay clean between nodes (only the bins a node filled are reset after it): most nodes
            // hold a handful of primitives, and resetting and sweeping all 3 x kBins bins for them was most of the build
            static thread_local Bins tl_bins;
            static thread_local bool tl_clean = false;
            if (!tl_clean) { tl_bins.reset(kMaxBins); tl_clean = true; }
            std::vector<Bins> many;
            if (parts > 1) { many.resize((size_t)parts - 1); for (Bins &B : many) B.reset(kBins); }
            Bins &A = tl_bins;
            split_run(b, e, parts, [&](int t, size_t pb, size_t pe) {
                Bins &B = t == 0 ? A : many[(size_t)t - 1];
                for (size_t i = pb; i < pe; ++i) {
                    const Prim &P = prims[i];
                    for (int ax = 0; ax < 3; ++ax) {
                        if (!use[ax]) continue;
                        int k = (int)((cen(P, ax) - clo[ax]) * scale3[ax]);
                        k = k < 0 ? 0 : (k >= kBins ? kBins - 1 : k);
                        B.cnt[ax][k]++;
                        box_grow(B.blo[ax][k], B.bhi[ax][k], P.lo, P.hi);
                    }
                }
            });
            for (const Bins &B : many)
                for (int ax = 0; ax < 3; ++ax)
                    for (int k = 0; k < kBins; ++k) {
                        A.cnt[ax][k] += B.cnt[ax][k];
                        box_grow(A.blo[ax][k], A.bhi[ax][k], B.blo[ax][k], B.bhi[ax][k]);
                    }
            for (int ax = 0; ax < 3; ++ax) {
                if (!use[ax]) continue;
                uint32_t *cnt = A.cnt[ax];
                float (*blo)[3] = A.blo[ax], (*bhi)[3] = A.bhi[ax];
                // only the non-empty bins matter: a split after an empty bin costs exactly what the split after the
                // last non-empty bin before it costs, and the first of equal costs is the one kept
                int nz[kMaxBins], m = 0;
                for (int k = 0; k < kBins; ++k) if (cnt[k]) nz[m++] = k;
                float ra[kMaxBins]; uint32_t rc[kMaxBins];     // area / count of non-empty bins j .. m-1
                float lo[3], hi[3];
                box_reset(lo, hi);
                uint32_t c = 0;
                for (int j = m - 1; j > 0; --j) {
                    box_grow(lo, hi, blo[nz[j]], bhi[nz[j]]);
                    c += cnt[nz[j]];
                    rc[j] = c; ra[j] = half_area(lo, hi);
                }
                box_reset(lo, hi);
                c = 0;
                for (int j = 0; j + 1 < m; ++j) {
                    box_grow(lo, hi, blo[nz[j]], bhi[nz[j]]);
                    c += cnt[nz[j]];
                    float cost = half_area(lo, hi) * (float)c + ra[j + 1] * (float)rc[j + 1];
                    if (cost < best_cost) { best_cost = cost; best_axis = ax; best_bin = nz[j]; }
                }
                for (int j = 0; j < m; ++j) { cnt[nz[j]] = 0; box_reset(blo[nz[j]], bhi[nz[j]]); }      // clean again
            }
        }
        if (extra) free_threads.fetch_add(extra);      // the subtrees below may take them
        size_t mid = b;
        if (best_axis >= 0) {
            float split_cost = kCostNode + kCostTri * best_cost / (parent_area > 0.0f ? parent_area : 1.0f);
            float leaf_cost = kCostTri * (float)n;
            if (n <= kMaxLeaf && leaf_cost <= split_cost) { make_leaf(); return; }
            float ext = chi[best_axis] - clo[best_axis];
            float scale = (float)kBins / ext;
            float c0 = clo[best_axis];
            int ax = best_axis, bin = best_bin;
            Prim *p = std::partition(prims.data() + b, prims.data() + e, [=](const Prim &P) {
                int k = (int)((cen(P, ax) - c0) * scale);
                k = k < 0 ? 0 : (k >= kBins ? kBins - 1 : k);
                return k <= bin;
            });
            mid = (size_t)(p - prims.data());
        }
        if (mid == b || mid == e) {
            // no usable SAH split (coincident centroids, or the depth guard): object median
            if (n <= kMaxLeaf) { make_leaf(); return; }
            int ax = 0;
            float ex = chi[0] - clo[0];
            for (int k = 1; k < 3; ++k) if (chi[k] - clo[k] > ex) { ex = chi[k] - clo[k]; ax = k; }
            mid = b + n / 2;
            std::nth_element(prims.begin() + b, prims.begin() + mid, prims.begin() + e,
                             [ax](const Prim &A, const Prim &B) {
                                 const float ca = cen(A, ax), cb = cen(B, ax);
                                 return ca < cb || (ca == cb && A.seq < B.seq);
                             });
        }
        uint32_t l = alloc(), r = alloc();
        N.left = (int32_t)l; N.right = (int32_t)r; N.first = 0; N.count = 0;
        size_t nl = mid - b;
        if (std::min(nl, n - nl) >= kSpawnMin && free_threads.fetch_sub(1) > 0) {
            // the helper hands its token back itself, as soon as it is done: joined only after this thread's own half
            std::thread t([this, l, b, mid, depth]() { build(l, b, mid, depth + 1); free_threads.fetch_add(1); });
            build(r, mid, e, depth + 1);
            t.join();
        } else {
            if (std::min(nl, n - nl) >= kSpawnMin) free_threads.fetch_add(1); // undo failed grab
            build(l, b, mid, depth + 1);
            build(r, mid, e, depth + 1);
        }
        nodes[ni].inner = 1u + nodes[l].inner + nodes[r].inner;
    }
};

inline int32_t leaf_code(uint32_t first, uint32_t count) { return ~(int32_t)((first << 3) | (count - 1)); }

}  // namespace

void rt_bvh_build(const rt_triangle *tris, size_t n_tris, const rt_mesh_info *meshes, size_t n_meshes,
                  const rt_vertex *verts, size_t n_verts, int n_threads, RtBvhBuildResult *out,
                  const rt_sphere *spheres, size_t n_spheres) {
    (void)n_verts;
    auto t0 = std::chrono::steady_clock::now();
    out->nodes.clear(); out->recs.clear();
    out->max_depth = 0; out->max_leaf = 0; out->sah_cost = 0.0; out->dropped = 0;

    if (const char *e = std::getenv("RT_TUNE_COST_TRI")) kCostTri = (float)std::atof(e);
    if (const char *e = std::getenv("RT_TUNE_BINS")) { int v = std::atoi(e); if (v >= 4 && v <= kMaxBins) kBins = v; }
    if (const char *e = std::getenv("RT_TUNE_MAX_LEAF")) { int v = std::atoi(e); if (v >= 1 && v <= 8) kMaxLeaf = (uint32_t)v; }
    if (n_threads <= 0) {
        n_threads = (int)std::thread::hardware_concurrency();
        if (n_threads > 16) n_threads = 16;
        if (n_threads < 1) n_threads = 1;
    }
    Builder B;
    // primitives in the reference's loop order (mesh by mesh, then the spheres): seq = position in that order
    std::vector<size_t> mesh_off(n_meshes + 1, 0);
    for (size_t m = 0; m < n_meshes; ++m) mesh_off[m + 1] = mesh_off[m] + meshes[m].tri_count;
    const size_t total = mesh_off[n_meshes];
    B.prims.resize(total + n_spheres);
    std::vector<uint8_t> keep;                    // only allocated if something has to be dropped
    std::atomic<size_t> dropped{0};
    std::atomic<bool> any_drop{false};
    auto bake_meshes = [&](int, size_t mb, size_t me) {
        size_t local_drop = 0;
        for (size_t m = mb; m < me; ++m) {
            size_t seq = mesh_off[m];
            for (uint32_t j = meshes[m].first_tri; j < meshes[m].first_tri + meshes[m].tri_count; ++j, ++seq) {
                const float *a = verts[tris[j].indices[0]].pos;
                const float *b = verts[tris[j].indices[1]].pos;
                const float *c = verts[tris[j].indices[2]].pos;
                Prim &P = B.prims[seq];
                bool finite = true;
                float amax = 0.0f;
                for (int k = 0; k < 3; ++k) {
                    finite = finite && std::isfinite(a[k]) && std::isfinite(b[k]) && std::isfinite(c[k]);
                    P.lo[k] = std::min(a[k], std::min(b[k], c[k]));
                    P.hi[k] = std::max(a[k], std::max(b[k], c[k]));
                    amax = std::max(amax, std::max(std::fabs(P.lo[k]), std::fabs(P.hi[k])));
                }
                // pad so that rounding in the slab test can never cull a triangle the exact f32
                // ray_triangle arithmetic accepts (DESIGN.md "Conservative culling")
                float pad = 1e-5f * amax + 1e-30f;
                for (int k = 0; k < 3; ++k) { P.lo[k] -= pad; P.hi[k] += pad; }
                P.seq = (uint32_t)seq; P.tri = j;
                // A triangle with a NaN/inf vertex can never pass ray_triangle's comparisons in a
                // way a box could bound; it is left out of the tree (marked here, removed below).
                if (!finite) { P.tri = kDroppedTri; ++local_drop; }
            }
        }
        if (local_drop) { dropped += local_drop; any_drop = true; }
    };
    Builder::split_run(0, n_meshes, total >= ((size_t)1 << 16) ? n_threads : 1, bake_meshes);
    for (size_t i = 0; i < n_spheres; ++i) {
        const rt_sphere &sp = spheres[i];
        Prim &P = B.prims[total + i];
        bool finite = std::isfinite(sp.radius);
        float amax = 0.0f;
        float r = std::fabs(sp.radius);
        for (int k = 0; k < 3; ++k) {
            finite = finite && std::isfinite(sp.center[k]);
            P.lo[k] = sp.center[k] - r; P.hi[k] = sp.center[k] + r;
            amax = std::max(amax, std::max(std::fabs(P.lo[k]), std::fabs(P.hi[k])));
        }
        // same padding rule as triangles, plus the rounding of centre +/- radius itself
        float pad = 2e-5f * amax + 1e-30f;
        for (int k = 0; k < 3; ++k) { P.lo[k] -= pad; P.hi[k] += pad; }
        P.seq = (uint32_t)(total + i); P.tri = (uint32_t)(n_tris + i);
        if (!finite) { P.tri = kDroppedTri; dropped += 1; any_drop = true; }
    }
    if (any_drop.load()) {          // rare: compact, order kept
        size_t w = 0;
        for (size_t i = 0; i < B.prims.size(); ++i) if (B.prims[i].tri != kDroppedTri) B.prims[w++] = B.prims[i];
        B.prims.resize(w);
        out->dropped = (uint32_t)dropped.load();
    }
    size_t n = B.prims.size();
    if (n == 0) { out->build_ms = 0.0; return; }

    B.free_threads = n_threads - 1;
    B.nodes.reset(new TmpNode[2 * n + 1]);
    uint32_t root = B.alloc();
    B.build(root, 0, n, 0);

    // leaf records in prim order
    out->recs.resize(n);
    Builder::split_run(0, n, n >= ((size_t)1 << 16) ? n_threads : 1, [&](int, size_t ib, size_t ie) {
        for (size_t i = ib; i < ie; ++i) {
            const Prim &P = B.prims[i];
            RtTriRecord &R = out->recs[i];
            if (P.seq >= total) {                 // the spheres come after every mesh in loop order
                const rt_sphere &sp = spheres[P.tri - n_tris];
                std::memset(&R, 0, sizeof R);
                R.a[0] = sp.center[0]; R.a[1] = sp.center[1]; R.a[2] = sp.center[2];
                R.abx = sp.radius;
                R.tri = P.tri; R.mesh = RT_REC_SPHERE; R.seq = P.seq;
                continue;
            }
            const float *a = verts[tris[P.tri].indices[0]].pos;
            const float *b = verts[tris[P.tri].indices[1]].pos;
            const float *c = verts[tris[P.tri].indices[2]].pos;
            R.a[0] = a[0]; R.a[1] = a[1]; R.a[2] = a[2];
            // edge_ab = b.pos - a.pos, edge_ac = c.pos - a.pos: plain f32 subtractions, the same
            // values ray_tracing.wgsl:122-123 computes per test
            R.abx = b[0] - a[0]; R.aby = b[1] - a[1]; R.abz = b[2] - a[2];
            R.acx = c[0] - a[0]; R.acy = c[1] - a[1]; R.acz = c[2] - a[2];
            // its mesh: the one whose run of the loop order holds seq
            R.tri = P.tri; R.seq = P.seq;
            R.mesh = (uint32_t)(std::upper_bound(mesh_off.begin(), mesh_off.end(), (size_t)P.seq) - mesh_off.begin() - 1);
        }
    });

    // flatten to the 64-B two-child layout: a node's inner children sit next to each other, then everything below
    // the second child, then everything below the first (the order a stack that pushes child 0, then child 1 gives).
    // With the subtree sizes from the build every node's place is known in advance, so subtrees flatten in parallel.
    const float qnan = std::numeric_limits<float>::quiet_NaN();
    const TmpNode &R = B.nodes[root];
    double root_area = half_area(R.lo, R.hi);
    if (!(root_area > 0.0)) root_area = 1.0;
    double sah = 0.0;
    uint32_t max_leaf = 0;
    if (R.left < 0) {
        RtBvhNode node;
        std::memset(&node, 0, sizeof node);
        for (int k = 0; k < 3; ++k) { node.lo0[k] = R.lo[k]; node.hi0[k] = R.hi[k]; node.lo1[k] = qnan; node.hi1[k] = qnan; }
        node.child0 = leaf_code(R.first, R.count);
        node.child1 = RT_CHILD_NONE;
        out->nodes.push_back(node);
        max_leaf = R.count;
        sah = kCostNode + kCostTri * R.count;
    } else {
        out->nodes.resize(R.inner);
        struct Item { uint32_t tmp, out, base; };      // base: where this node's inner children go
        B.free_threads = n_threads - 1;
        std::atomic<uint32_t> max_leaf_all{0};
        std::function<void(Item)> flatten = [&](Item first) {
            std::vector<Item> stack{first};
            std::vector<std::thread> helpers;
            uint32_t ml = 0;
            while (!stack.empty()) {
                Item it = stack.back(); stack.pop_back();
                const TmpNode &T = B.nodes[it.tmp];
                const TmpNode *ch[2] = {&B.nodes[T.left], &B.nodes[T.right]};
                RtBvhNode node;
                std::memset(&node, 0, sizeof node);
                int32_t code[2];
                uint32_t next = it.base;
                for (int c = 0; c < 2; ++c) {
                    if (ch[c]->left < 0) { code[c] = leaf_code(ch[c]->first, ch[c]->count); ml = std::max(ml, ch[c]->count); }
                    else code[c] = (int32_t)next++;
                }
                for (int k = 0; k < 3; ++k) {
                    node.lo0[k] = ch[0]->lo[k]; node.hi0[k] = ch[0]->hi[k];
                    node.lo1[k] = ch[1]->lo[k]; node.hi1[k] = ch[1]->hi[k];
                }
                node.child0 = code[0]; node.child1 = code[1];
                out->nodes[it.out] = node;
                // below child 1 first, then below child 0
                uint32_t base1 = next, base0 = next + (ch[1]->left < 0 ? 0u : ch[1]->inner - 1u);
                if (ch[0]->left >= 0) {
                    Item i0{(uint32_t)T.left, (uint32_t)code[0], base0};
                    if (ch[0]->inner >= (1u << 14) && B.free_threads.fetch_sub(1) > 0)
                        helpers.emplace_back([&flatten, &B, i0]() { flatten(i0); B.free_threads.fetch_add(1); });
                    else { if (ch[0]->inner >= (1u << 14)) B.free_threads.fetch_add(1); stack.push_back(i0); }
                }
                if (ch[1]->left >= 0) stack.push_back({(uint32_t)T.right, (uint32_t)code[1], base1});
            }
            for (auto &t : helpers) t.join();
            uint32_t cur = max_leaf_all.load();
            while (ml > cur && !max_leaf_all.compare_exchange_weak(cur, ml)) {}
        };
        flatten({root, 0u, 1u});
        max_leaf = max_leaf_all.load();
        // SAH cost of the tree, summed in storage order
        for (const RtBvhNode &N : out->nodes) {
            float lo[3], hi[3];
            box_reset(lo, hi);
            box_grow(lo, hi, N.lo0, N.hi0); box_grow(lo, hi, N.lo1, N.hi1);
            sah += kCostNode * half_area(lo, hi) / root_area;
            if (N.child0 < 0) sah += kCostTri * ((((uint32_t)~N.child0) & 7u) + 1u) * half_area(N.lo0, N.hi0) / root_area;
            if (N.child1 < 0) sah += kCostTri * ((((uint32_t)~N.child1) & 7u) + 1u) * half_area(N.lo1, N.hi1) / root_area;
        }
    }
    out->max_depth = B.max_depth.load();
    out->max_leaf = max_leaf;
    out->sah_cost = sah;
    out->build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

bool rt_bvh_refit(RtBvhBuildResult *bvh, const rt_triangle *tris, const rt_vertex *verts, int n_threads,
                  const rt_sphere *spheres, size_t n_tris) {
    (void)spheres; (void)n_tris;
    auto t0 = std::chrono::steady_clock::now();
    const size_t n = bvh->recs.size();
    if (n == 0) return true;
    if (n_threads <= 0) {
        n_threads = (int)std::thread::hardware_concurrency();
        if (n_threads > 16) n_threads = 16;
        if (n_threads < 1) n_threads = 1;
    }
    if (n < (size_t)1 << 16) n_threads = 1;
    std::vector<float> blo(3 * n), bhi(3 * n);
    std::atomic<bool> finite{true};
    auto bake = [&](size_t b, size_t e) {
        for (size_t i = b; i < e; ++i) {
            RtTriRecord &R = bvh->recs[i];
            if (R.mesh == RT_REC_SPHERE) {     // spheres do not move in a vertex update: same box as at build time
                float r = std::fabs(R.abx), amax = 0.0f;
                for (int k = 0; k < 3; ++k) amax = std::max(amax, std::max(std::fabs(R.a[k] - r), std::fabs(R.a[k] + r)));
                float pad = 2e-5f * amax + 1e-30f;
                for (int k = 0; k < 3; ++k) { blo[3 * i + k] = R.a[k] - r - pad; bhi[3 * i + k] = R.a[k] + r + pad; }
                continue;
            }
            const float *a = verts[tris[R.tri].indices[0]].pos;
            const float *bb = verts[tris[R.tri].indices[1]].pos;
            const float *c = verts[tris[R.tri].indices[2]].pos;
            float amax = 0.0f, lo[3], hi[3];
            bool ok = true;
            for (int k = 0; k < 3; ++k) {
                ok = ok && std::isfinite(a[k]) && std::isfinite(bb[k]) && std::isfinite(c[k]);
                lo[k] = std::min(a[k], std::min(bb[k], c[k]));
                hi[k] = std::max(a[k], std::max(bb[k], c[k]));
                amax = std::max(amax, std::max(std::fabs(lo[k]), std::fabs(hi[k])));
            }
            if (!ok) { finite = false; return; }
            float pad = 1e-5f * amax + 1e-30f;    // same padding rule as the build
            for (int k = 0; k < 3; ++k) { blo[3 * i + k] = lo[k] - pad; bhi[3 * i + k] = hi[k] + pad; }
            R.a[0] = a[0]; R.a[1] = a[1]; R.a[2] = a[2];
            R.abx = bb[0] - a[0]; R.aby = bb[1] - a[1]; R.abz = bb[2] - a[2];
            R.acx = c[0] - a[0]; R.acy = c[1] - a[1]; R.acz = c[2] - a[2];
        }
    };
    if (n_threads == 1) bake(0, n);
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < n_threads; ++t) th.emplace_back(bake, n * t / n_threads, n * (t + 1) / n_threads);
        for (auto &t : th) t.join();
    }
    if (!finite) return false;
    auto child_box = [&](int32_t code, float *lo, float *hi) {
        box_reset(lo, hi);
        if (code >= 0) {
            const RtBvhNode &C = bvh->nodes[code];
            if (C.child0 != RT_CHILD_NONE) box_grow(lo, hi, C.lo0, C.hi0);
            if (C.child1 != RT_CHILD_NONE) box_grow(lo, hi, C.lo1, C.hi1);
        } else {
            uint32_t v = (uint32_t)(~code), first = v >> 3, cnt = (v & 7u) + 1u;
            for (uint32_t k = first; k < first + cnt; ++k) box_grow(lo, hi, &blo[3 * k], &bhi[3 * k]);
        }
    };
    for (size_t i = bvh->nodes.size(); i-- > 0;) {     // children have larger indices than their parent
        RtBvhNode &N = bvh->nodes[i];
        if (N.child0 != RT_CHILD_NONE) child_box(N.child0, N.lo0, N.hi0);
        if (N.child1 != RT_CHILD_NONE) child_box(N.child1, N.lo1, N.hi1);
    }
    bvh->build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return true;
}

// Experiment (RT_TUNE_WIDE_ORDER, round 3): the wide nodes' ORDER IN MEMORY, a pure re-indexing -- the
// tree, its boxes, every ray's sequence of visits and the results stay what they are.  The walk in rt_bvh_widen already lays
// the children of a node out side by side and continues with the subtree of the last child (mode 0).
//   1: breadth-first over the whole tree (children still side by side; a level is one contiguous block)
//   2: breadth-first for the top of the tree (until a level holds >= 4096 nodes), the subtrees below it one after the other
//      in mode 0's order -- the "hot block + treelets" layout
//   3: a random permutation (root first): what is locality worth at all?
//   4: mode 0 with padding nodes, so that a group of 2 or 4 siblings never straddles one more 128-B line than it has to
static void wide_reorder(RtWideTree *w, int mode) {
    const size_t n = w->nodes.size();
    if (n < 2 || mode <= 0 || mode > 4) return;
    std::vector<uint32_t> order;            // order[new index] = old index; 0xFFFFFFFF = a padding node
    order.reserve(n + n / 4);
    auto children = [&](uint32_t i, uint32_t *c) {
        int m = 0;
        for (int k = 0; k < 4; ++k) if (w->nodes[i].child[k] >= 0) c[m++] = (uint32_t)w->nodes[i].child[k];
        return m;
    };
    auto cluster_dfs = [&](uint32_t root_old, bool emit_root) {     // mode 0's order below root_old
        std::vector<uint32_t> st;
        if (emit_root) order.push_back(root_old);
        st.push_back(root_old);
        while (!st.empty()) {
            uint32_t i = st.back(); st.pop_back();
            uint32_t c[4]; int m = children(i, c);
            if (mode == 4 && (m == 2 || m == 4) && (order.size() & 1u)) order.push_back(0xFFFFFFFFu);
            for (int k = 0; k < m; ++k) { order.push_back(c[k]); st.push_back(c[k]); }
        }
    };
    if (mode == 1 || mode == 2) {
        std::vector<uint32_t> level{0u}, next;
        order.push_back(0u);
        while (!level.empty()) {
            next.clear();
            for (uint32_t i : level) { uint32_t c[4]; int m = children(i, c); for (int k = 0; k < m; ++k) { order.push_back(c[k]); next.push_back(c[k]); } }
            if (mode == 2 && next.size() >= 4096) { for (uint32_t i : next) cluster_dfs(i, false); break; }
            level.swap(next);
        }
    } else if (mode == 3) {
        for (uint32_t i = 0; i < n; ++i) order.push_back(i);
        uint64_t x = 0x9E3779B97F4A7C15ull;
        for (size_t i = n - 1; i > 1; --i) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            std::swap(order[i], order[1 + (size_t)(x % i)]);
        }
    } else cluster_dfs(0u, true);
    std::vector<uint32_t> newidx(n, 0xFFFFFFFFu);
    for (size_t i = 0; i < order.size(); ++i) if (order[i] != 0xFFFFFFFFu) newidx[order[i]] = (uint32_t)i;
    std::vector<RtWideNode> nodes(order.size());
    std::vector<RtWideSrc> src(order.size());
    for (size_t i = 0; i < order.size(); ++i) {
        RtWideNode W; RtWideSrc S;
        if (order[i] == 0xFFFFFFFFu) {           // padding: an empty node nobody points at
            std::memset(&W, 0, sizeof W);
            for (int k = 0; k < 4; ++k) { W.child[k] = RT_CHILD_NONE; S.slot[k] = RT_WIDE_EMPTY; }
        } else {
            W = w->nodes[order[i]]; S = w->src[order[i]];
            for (int k = 0; k < 4; ++k) if (W.child[k] >= 0) W.child[k] = (int32_t)newidx[W.child[k]];
        }
        nodes[i] = W; src[i] = S;
    }
    w->nodes.swap(nodes); w->src.swap(src);
}

void rt_bvh_widen(const RtBvhBuildResult &bvh, RtWideTree *out) {
    auto t0 = std::chrono::steady_clock::now();
    out->nodes.clear(); out->src.clear(); out->max_depth = 0; out->stack_need = 0;
    if (bvh.nodes.empty()) { out->build_ms = 0.0; return; }
    struct Slot { int32_t code; uint32_t src; float area; };
    struct Item { uint32_t bin, wide, depth; };
    auto slots_of = [&](uint32_t n, Slot *s) {
        const RtBvhNode &N = bvh.nodes[n];
        int c = 0;
        if (N.child0 != RT_CHILD_NONE) s[c++] = {N.child0, (n << 1) | 0u, half_area(N.lo0, N.hi0)};
        if (N.child1 != RT_CHILD_NONE) s[c++] = {N.child1, (n << 1) | 1u, half_area(N.lo1, N.hi1)};
        return c;
    };
    out->nodes.reserve(bvh.nodes.size() / 2 + 1);
    out->src.reserve(bvh.nodes.size() / 2 + 1);
    std::vector<Item> stack;
    out->nodes.emplace_back(); out->src.emplace_back();
    stack.push_back({0u, 0u, 0u});
    while (!stack.empty()) {
        Item it = stack.back(); stack.pop_back();
        out->max_depth = std::max(out->max_depth, it.depth);
        Slot s[4];
        int n = slots_of(it.bin, s);
        while (n < 4) {
            int best = -1;
            for (int k = 0; k < n; ++k)
                if (s[k].code >= 0 && (best < 0 || s[k].area > s[best].area)) best = k;
            if (best < 0) break;
            Slot two[2];
            int m = slots_of((uint32_t)s[best].code, two);
            if (n - 1 + m > 4) break;
            // the two children take the place of their parent, order kept
            for (int k = n - 1; k > best; --k) s[k + m - 1] = s[k];
            for (int k = 0; k < m; ++k) s[best + k] = two[k];
            n += m - 1;
            if (m == 0) break;   // cannot happen (an inner node has at least one child); guards the loop
        }
        RtWideNode W;
        RtWideSrc S;
        std::memset(&W, 0, sizeof W);
        for (int k = 0; k < 4; ++k) {
            if (k >= n) { W.child[k] = RT_CHILD_NONE; S.slot[k] = RT_WIDE_EMPTY; continue; }
            S.slot[k] = s[k].src;
            if (s[k].code >= 0) {
                uint32_t idx = (uint32_t)out->nodes.size();
                out->nodes.emplace_back(); out->src.emplace_back();
                W.child[k] = (int32_t)idx;
                stack.push_back({(uint32_t)s[k].code, idx, it.depth + 1});
            } else W.child[k] = s[k].code;
        }
        out->nodes[it.wide] = W; out->src[it.wide] = S;
    }
    if (const char *e = std::getenv("RT_TUNE_WIDE_ORDER")) wide_reorder(out, std::atoi(e));   // experiments: the nodes' order in memory
    rt_wide_requantise(bvh, out);       // the boxes, on all threads (the walk above only fixes topology and order)
    // worst-case stack use: with n children hit, n-1 wait on the stack while the first subtree is walked, n-2
    // during the second, ...; the worst order visits the neediest subtree first
    std::vector<uint32_t> need(out->nodes.size(), 0);
    for (size_t i = out->nodes.size(); i-- > 0;) {
        uint32_t sub[4]; int n = 0;
        for (int k = 0; k < 4; ++k) {
            int32_t c = out->nodes[i].child[k];
            if (c == RT_CHILD_NONE) continue;
            sub[n++] = c >= 0 ? need[c] : 0u;
        }
        std::sort(sub, sub + n, [](uint32_t a, uint32_t b) { return a > b; });
        uint32_t w = 0;
        for (int k = 0; k < n; ++k) w = std::max(w, sub[k] + (uint32_t)(n - 1 - k));
        need[i] = w;
    }
    out->stack_need = need[0];
    out->build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

void rt_wide_requantise(const RtBvhBuildResult &bvh, RtWideTree *w) {
    const size_t n = w->nodes.size();
    int n_threads = (int)std::thread::hardware_concurrency();
    n_threads = n < ((size_t)1 << 16) ? 1 : (n_threads > 16 ? 16 : (n_threads < 1 ? 1 : n_threads));
    auto part = [&](size_t b, size_t e) {
        for (size_t i = b; i < e; ++i) rt_wide_quantise(bvh.nodes.data(), w->src[i], w->nodes[i]);
    };
    if (n_threads == 1) { part(0, n); return; }
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; ++t) th.emplace_back(part, n * t / n_threads, n * (t + 1) / n_threads);
    for (auto &t : th) t.join();
}
