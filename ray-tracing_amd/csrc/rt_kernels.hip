// rt_kernels.hip -- hand-written HIP kernels for gfx950 (MI355X / CDNA4): the per-pixel hot
// path of camms205/ray-tracing (assets/shaders/ray_tracing.wgsl:118-211), one thread per ray,
// wave64 = one 8x8 pixel region, traversal of a 4-wide quantised BVH (or the binary tree it was
// collapsed from) with an LDS-resident per-lane stack; persistent waves fed from per-XCD queues.
//
// Float contract (must match oracle/rt_oracle.c bit for bit; built with -ffp-contract=off):
//   dot = (x*x' + y*y') + z*z' ; cross component = p*q - r*s ; normalize = v / sqrt(dot(v,v));
//   '/' and sqrtf are correctly rounded (hipcc default); min/max = fminf/fmaxf.
#include <hip/hip_runtime.h>
#include "rt_kernels.h"
#include "rt_wide.h"

// Per-lane traversal stack: the first RT_STACK_LDS entries live in LDS (stack[level][thread], bank
// = lane: conflict-free), deeper levels spill to a private array.  16 entries x 256 threads = 16 KiB
// per block.  Every traversal kernel is built for 7 waves per SIMD (<= 72 VGPRs): 8 waves / 64 VGPRs spill more, 6 waves hide
// less latency (same-box A/Bs with grids sized to match: k_stream below; path kernels 7 vs 8 waves: 4K frame -0.9 %, 10 M
// triangles -2.1 %, 4 spp at 1080p -2.3 %; 6 waves +4 % / +3 % / +1.5 %).
#ifndef RT_STACK_LDS
#define RT_STACK_LDS 16
#endif
#ifndef RT_MIN_WAVES
#define RT_MIN_WAVES 7
#endif
// private part: a binary tree needs <= RT_MAX_DEPTH entries in all, a 4-wide tree up to 3 per level (the host
// checks RtWideTree::stack_need against RT_STACK_CAP before it enables the wide tree)
#define RT_STACK_OVF (RT_STACK_CAP - RT_STACK_LDS)
#define RT_INF __builtin_huge_valf()

struct f3 { float x, y, z; };
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f3 F3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ f3 sub(f3 a, f3 b) { return F3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ f3 add(f3 a, f3 b) { return F3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ f3 mul(f3 a, f3 b) { return F3(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ f3 scl(f3 a, float s) { return F3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ float dot(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ f3 cross(f3 a, f3 b) {
    return F3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ f3 normalize(f3 a) {
    float len = sqrtf(dot(a, a));
    return F3(a.x / len, a.y / len, a.z / len);
}

// ---- A1 ray generation: ray_tracing.wgsl:174,176-177 ---------------------------------------
__device__ __forceinline__ void ray_gen(const RtViewDev &V, uint32_t px, uint32_t py, f3 &o, f3 &d) {
    float u = ((float)(px - V.vx) + 0.5f) / (float)V.vw;
    float w = ((float)(py - V.vy) + 0.5f) / (float)V.vh;
    float nx = u * 2.0f + -1.0f;
    float ny = w * -2.0f + 1.0f;
    float r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
        r[i] = ((V.m[0 + i] * nx + V.m[4 + i] * ny) + V.m[8 + i] * 0.0f) + V.m[12 + i] * 1.0f;
    float len = sqrtf(((r[0] * r[0] + r[1] * r[1]) + r[2] * r[2]) + r[3] * r[3]);
    d = F3(r[0] / len, r[1] / len, r[2] / len);
    o = F3(V.pos[0], V.pos[1], V.pos[2]);
}

// ---- A3 ray/AABB cull: ray_tracing.wgsl:141-148 (inv = 1.0/dir, computed once per ray) --------
__device__ __forceinline__ bool ray_aabb_ref(f3 o, f3 inv, f3 lb, f3 rt) {
    f3 t1 = mul(sub(lb, o), inv);
    f3 t2 = mul(sub(rt, o), inv);
    float tmin = fmaxf(fmaxf(fminf(t1.x, t2.x), fminf(t1.y, t2.y)), fminf(t1.z, t2.z));
    float tmax = fminf(fminf(fmaxf(t1.x, t2.x), fmaxf(t1.y, t2.y)), fmaxf(t1.z, t2.z));
    return tmax >= tmin;
}

// ---- A4 ray/triangle: ray_tracing.wgsl:118-139, from a, ab, ac ----------------------------------
__device__ __forceinline__ bool ray_tri(f3 o, f3 d, f3 a, f3 ab, f3 ac, float &dst, float &u, float &v) {
    f3 norm = cross(ab, ac);
    f3 ao = sub(o, a);
    f3 dao = cross(ao, d);
    float det = -dot(d, norm);
    float inv_det = 1.0f / det;
    dst = dot(ao, norm) * inv_det;
    u = dot(ac, dao) * inv_det;
    v = -dot(ab, dao) * inv_det;
    float w = 1.0f - u - v;
    return det >= 1e-6f && dst >= 0.0f && u >= 0.0f && v >= 0.0f && w >= 0.0f;
}

// A ray with a NaN in its origin or direction misses everything in the reference's loop (every compare of wgsl:137 is
// false; a sphere's discriminant is NaN).  It is ended before it starts: the quantised wide step drops NaN operands
// through min/max and would walk the WHOLE tree for it (a multi-second kernel on the big scenes).  Such rays
// come from zero-length vertex normals (shadow-ray origin) or a degenerate world_from_clip (0/0 directions).
__device__ __forceinline__ bool ray_has_nan(f3 o, f3 d) {
    return o.x != o.x || o.y != o.y || o.z != o.z || d.x != d.x || d.y != d.y || d.z != d.z;
}

struct q4 { float x, y, z, w; };
// ---- extension: ray / analytic sphere (build-defined; oracle orc_ray_sphere, same operation order) ----------
__device__ __forceinline__ bool ray_sphere(f3 o, f3 d, f3 c, float r, float &dst) {
    f3 oc = sub(o, c);
    float a = dot(d, d);
    float hb = dot(oc, d);
    float cc = dot(oc, oc) - r * r;
    float disc = hb * hb - a * cc;
    dst = (-hb - sqrtf(disc)) / a;
    return disc >= 0.0f && dst >= 0.0f;      // entry point only: an origin inside the sphere misses it
}
// one 48-B leaf record against the ray: a triangle (a, ab, ac) or a sphere (centre, radius; mesh = RT_REC_SPHERE)
template <class Q>   // Q: float4, or the broadcast q4 of the wave-coherent traversal
__device__ __forceinline__ bool prim_test(f3 o, f3 d, Q r0, Q r1, Q r2, float &dst, float &u, float &v) {
    if (__float_as_uint(r2.z) == RT_REC_SPHERE) {
        u = 0.0f; v = 0.0f;
        return ray_sphere(o, d, F3(r0.x, r0.y, r0.z), r0.w, dst);
    }
    return ray_tri(o, d, F3(r0.x, r0.y, r0.z), F3(r0.w, r1.x, r1.y), F3(r1.z, r1.w, r2.x), dst, u, v);
}

struct Hit {
    float t, u, v;
    uint32_t seq, tri, mesh;
    bool hit;
};

struct Cnt { uint32_t nodes, tris, meshes, wnodes, wtris, walive; };   // w*: wave-level loop trips (first active lane counts)

#define RT_DONE ((int)0x80000001)
// The lane's stack: entries 0..RT_STACK_LDS-1 in its LDS column, deeper ones in a private array.  Both
// are reached through address-space-qualified pointers so that the accesses are ds_read/ds_write and
// scratch_load/scratch_store (a generic pointer -- or a select between the two -- becomes a flat access).
typedef __attribute__((address_space(3))) int lds_int;
typedef __attribute__((address_space(5))) int prv_int;
__device__ __forceinline__ int stack_pop(int &sp, const lds_int *stk, const prv_int *ovf) {
    if (sp == 0) return RT_DONE;
    --sp;
    int v = stk[(sp < RT_STACK_LDS ? sp : RT_STACK_LDS - 1) * RT_BLOCK];
    if (sp >= RT_STACK_LDS) v = ovf[sp - RT_STACK_LDS];
    return v;
}
// Pop with the LDS part's top entry already in a register (read at the start of the trip, when the trip's loads were issued:
// a step pops only if it pushed nothing, so the entry is still the top): the LDS round trip leaves the trip's critical path.
__device__ __forceinline__ int stack_pop_pref(int &sp, const prv_int *ovf, int tos_pref) {
    if (sp == 0) return RT_DONE;
    --sp;
    int v = tos_pref;
    if (sp >= RT_STACK_LDS) v = ovf[sp - RT_STACK_LDS];
    return v;
}
__device__ __forceinline__ int stack_peek_lds(int sp, const lds_int *stk) {
    const int i = sp > 0 ? sp - 1 : 0;
    return stk[(i < RT_STACK_LDS ? i : RT_STACK_LDS - 1) * RT_BLOCK];
}
__device__ __forceinline__ void stack_push(int &sp, lds_int *stk, prv_int *ovf, int v) {
    if (sp < RT_STACK_LDS) stk[sp * RT_BLOCK] = v;
    else if (sp - RT_STACK_LDS < RT_STACK_OVF) ovf[sp - RT_STACK_LDS] = v;
    ++sp;
}


// One node step: the two child boxes of the fetched node against the lane's ray, then descend / push / pop.
// Q: float4 (per-lane vector fetch) or q4 (one scalar fetch for the whole wave, operands in SGPRs).
template <bool PREF, class Q>
__device__ __forceinline__ void node_decide(Q q0, Q q1, Q q2, Q q3, f3 o, f3 inv, float t, int &cur, int &sp, lds_int *stk,
                                            prv_int *ovf, int tos_pref) {
    const float limit = t * 1.00001f;   // relaxed bound: near-ties are still visited, seq decides
    float ax0 = (q0.x - o.x) * inv.x, ax1 = (q0.w - o.x) * inv.x;
    float ay0 = (q0.y - o.y) * inv.y, ay1 = (q1.x - o.y) * inv.y;
    float az0 = (q0.z - o.z) * inv.z, az1 = (q1.y - o.z) * inv.z;
    float bx0 = (q1.z - o.x) * inv.x, bx1 = (q2.y - o.x) * inv.x;
    float by0 = (q1.w - o.y) * inv.y, by1 = (q2.z - o.y) * inv.y;
    float bz0 = (q2.x - o.z) * inv.z, bz1 = (q2.w - o.z) * inv.z;
    float tmin0 = fmaxf(fmaxf(fminf(ax0, ax1), fminf(ay0, ay1)), fminf(az0, az1));
    float tmax0 = fminf(fminf(fmaxf(ax0, ax1), fmaxf(ay0, ay1)), fmaxf(az0, az1));
    float tmin1 = fmaxf(fmaxf(fminf(bx0, bx1), fminf(by0, by1)), fminf(bz0, bz1));
    float tmax1 = fminf(fminf(fmaxf(bx0, bx1), fmaxf(by0, by1)), fmaxf(bz0, bz1));
    tmax0 *= 1.0000005f; tmax1 *= 1.0000005f;   // robust slab test (2 ulp)
    bool h0 = tmin0 <= tmax0 && tmax0 >= 0.0f && tmin0 <= limit;
    bool h1 = tmin1 <= tmax1 && tmax1 >= 0.0f && tmin1 <= limit;
    int c0 = __float_as_int(q3.x), c1 = __float_as_int(q3.y);
    bool both = h0 && h1;
    bool swap = tmin1 < tmin0;
    if (both) stack_push(sp, stk, ovf, swap ? c0 : c1);
    if (h0 || h1) cur = (both ? swap : h1) ? c1 : c0;
    else cur = PREF ? stack_pop_pref(sp, ovf, tos_pref) : stack_pop(sp, stk, ovf);
}

// One step on the 4-wide tree: four quantised child boxes against the lane's ray, the hit ones sorted by entry
// distance, nearest taken, the others pushed farthest first.  Plane = org + byte*scl, so with A = scl*inv and
// B = (o - org)*inv the slab distance is ONE fma per plane: byte*A - B (NaN / inf from a zero direction component
// are dropped by min/max, i.e. the axis does not constrain -- conservative).  The boxes contain the binary tree's
// (rt_wide.h), the far distance gets 8 ulp of relative slack for the rounding of this form with a distant origin.
// The wide step's  byte*A - B  form cannot use an infinite 1/d (inf - inf): a direction component of (nearly) zero
// gets +-1e30 instead -- the ray then drifts 1e-30 per unit of t on that axis, far inside the 1e-12 every box is
// padded with (rt_wide.h).  The per-mesh cull of wgsl:155 still takes the exact 1/d.
#define RT_WIDE_INV_MAX 1e30f
template <bool WIDE>
__device__ __forceinline__ f3 ray_inv(f3 d) {
    f3 inv = F3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    if (WIDE) inv = F3(fminf(fmaxf(inv.x, -RT_WIDE_INV_MAX), RT_WIDE_INV_MAX), fminf(fmaxf(inv.y, -RT_WIDE_INV_MAX), RT_WIDE_INV_MAX),
                       fminf(fmaxf(inv.z, -RT_WIDE_INV_MAX), RT_WIDE_INV_MAX));
    return inv;
}
// The exact 1/d for the per-mesh cull, from the clamped one: a component that was not clamped IS 1/d; only a clamped
// one (a direction component of (nearly) zero) is divided again.  Keeps three registers out of the traversal loop.
template <bool WIDE>
__device__ __forceinline__ f3 exact_inv(f3 inv, f3 d) {
    if (WIDE) {
        if (!(fabsf(inv.x) < RT_WIDE_INV_MAX)) inv.x = 1.0f / d.x;
        if (!(fabsf(inv.y) < RT_WIDE_INV_MAX)) inv.y = 1.0f / d.y;
        if (!(fabsf(inv.z) < RT_WIDE_INV_MAX)) inv.z = 1.0f / d.z;
    }
    return inv;
}
// ---- wave priorities (round 3) ------------------------------------------------------------------------------------
// A trip of the traversal loop is: vote -> addresses -> the node's (record's) loads -> wait -> ~125 (~60) VALU instructions of
// slab tests, ordering and pushes -> next trip.  With eight waves on a SIMD the arbiter picks the oldest ready wave, so a wave
// that is ten instructions away from issuing its next loads -- one or two microseconds of memory latency that nothing else can
// start -- queues behind waves that have a hundred instructions of arithmetic ahead of them.  s_setprio (0..3, per wave, among
// the waves of its SIMD) turns that into shortest-time-to-next-load-first: the priority RISES through the step and is dropped
// as soon as the loads are out.  Same instructions, same results; measured (profiles/r3/ab_setprio*.txt): bench scene -5.6 %
// one frame at a time, -6.4 % with four in flight; 1 M-triangle soup -4.5 % / -5 %; path tracing 10 M triangles +3 %, 4K +5 ... +7 %.
#define RT_PRIO_ISSUE 3     // from the vote to the issue of the trip's loads; the hand-over (service) code, which ends in new rays
#define RT_PRIO_WAIT 0      // loads issued: wait for them, start the step's arithmetic
#define RT_PRIO_ORDER 1     // wide step: slab tests done, ordering the hit children
#define RT_PRIO_PUSH 2      // wide step: children ordered, pushes and the next node left; record step: primitive test done
#define RT_SETPRIO(p) __builtin_amdgcn_s_setprio(p)
__device__ __forceinline__ float ub(uint32_t w, int k) { return (float)((w >> (8 * k)) & 0xFFu); }   // v_cvt_f32_ubyteK
template <bool PREF, class Q>
__device__ __forceinline__ void wide_decide(Q q0, Q q1, Q q2, Q q3, f3 o, f3 inv, float t, int &cur, int &sp, lds_int *stk,
                                            prv_int *ovf, int tos_pref) {
    const float limit = t * 1.00001f;   // relaxed bound: near-ties are still visited, seq decides
    // plane = org + byte*scl  =>  distance = byte*A - B with A = scl*inv, B = (o - org)*inv: one fma per plane.
    // (Two children per v_pk_fma_f32 was measured: 1 % slower -- packed f32 issues at half rate here.)
    const float Ax = q0.w * inv.x, Ay = q1.x * inv.y, Az = q1.y * inv.z;
    const float Bx = (o.x - q0.x) * inv.x, By = (o.y - q0.y) * inv.y, Bz = (o.z - q0.z) * inv.z;
    const uint32_t lox = __float_as_uint(q1.z), loy = __float_as_uint(q1.w), loz = __float_as_uint(q2.x);
    const uint32_t hix = __float_as_uint(q2.y), hiy = __float_as_uint(q2.z), hiz = __float_as_uint(q2.w);
    const bool sx = inv.x < 0.0f, sy = inv.y < 0.0f, sz = inv.z < 0.0f;
    const uint32_t nx = sx ? hix : lox, fx = sx ? lox : hix;
    const uint32_t ny = sy ? hiy : loy, fy = sy ? loy : hiy;
    const uint32_t nz = sz ? hiz : loz, fz = sz ? loz : hiz;
    uint32_t key[4]; int ref[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float tn = fmaxf(fmaxf(__builtin_fmaf(ub(nx, k), Ax, -Bx), __builtin_fmaf(ub(ny, k), Ay, -By)), __builtin_fmaf(ub(nz, k), Az, -Bz));
        float tf = fminf(fminf(__builtin_fmaf(ub(fx, k), Ax, -Bx), __builtin_fmaf(ub(fy, k), Ay, -By)), __builtin_fmaf(ub(fz, k), Az, -Bz));
        tn = fmaxf(tn, 0.0f);
        tf = fminf(tf * 1.000001f, limit);
        const int c = __float_as_int(k == 0 ? q3.x : (k == 1 ? q3.y : (k == 2 ? q3.z : q3.w)));
        // an empty slot has child = RT_CHILD_NONE and an inverted box; should rounding ever let that box "hit", the
        // slot only sorts among the hits and is skipped below like a miss
        const bool hit = tn <= tf;
        key[k] = hit ? __float_as_uint(tn) : 0xFFFFFFFFu;      // tn >= 0: its bits order like the value
        ref[k] = hit ? c : RT_CHILD_NONE;
    }
#define RT_CE(a, b) { const bool s_ = key[b] < key[a]; const uint32_t ka = s_ ? key[b] : key[a], kb = s_ ? key[a] : key[b]; \
                      const int ra = s_ ? ref[b] : ref[a], rb = s_ ? ref[a] : ref[b]; key[a] = ka; key[b] = kb; ref[a] = ra; ref[b] = rb; }
    // nearest child exactly, the rest roughly: three exchanges put the minimum first and leave (max01, second
    // minimum, max23) behind it; taking the second minimum next is right most of the time.  The full five-exchange
    // sort visits 1-3 % fewer nodes and is 0-2 % slower (10 VALU instructions more per step).
    RT_SETPRIO(RT_PRIO_ORDER);
    RT_CE(0, 1) RT_CE(2, 3) RT_CE(0, 2)
    { const uint32_t tk = key[1]; const int tr = ref[1]; key[1] = key[2]; ref[1] = ref[2]; key[2] = tk; ref[2] = tr; }
#undef RT_CE
    RT_SETPRIO(RT_PRIO_PUSH);
    if (sp + 3 <= RT_STACK_LDS) {      // the usual case: three slots left in the LDS part
        if (ref[3] != RT_CHILD_NONE) { stk[sp * RT_BLOCK] = ref[3]; ++sp; }
        if (ref[2] != RT_CHILD_NONE) { stk[sp * RT_BLOCK] = ref[2]; ++sp; }
        if (ref[1] != RT_CHILD_NONE) { stk[sp * RT_BLOCK] = ref[1]; ++sp; }
    } else {
        if (ref[3] != RT_CHILD_NONE) stack_push(sp, stk, ovf, ref[3]);
        if (ref[2] != RT_CHILD_NONE) stack_push(sp, stk, ovf, ref[2]);
        if (ref[1] != RT_CHILD_NONE) stack_push(sp, stk, ovf, ref[1]);
    }
#ifdef RT_EXP_EXTRA_VALU      // experiment only: N more VALU instructions per wide step (is the loaded phase VALU-bound?)
    {
        float dmy = Ax;
#pragma unroll
        for (int i = 0; i < RT_EXP_EXTRA_VALU; ++i) dmy = __builtin_fmaf(dmy, Ay, Az);
        if (dmy == 12345.678f) ref[0] = RT_CHILD_NONE;
    }
#endif
#ifdef RT_EXP_EXTRA_VALU_SHADOW   // experiment only: N more VALU instructions in the steps of +Y shadow rays alone (what would a cheaper shadow step buy?)
    if (inv.x == RT_WIDE_INV_MAX && inv.z == RT_WIDE_INV_MAX) {
        float dmy = Ax;
#pragma unroll
        for (int i = 0; i < RT_EXP_EXTRA_VALU_SHADOW; ++i) dmy = __builtin_fmaf(dmy, Ay, Az);
        if (dmy == 12345.678f) ref[0] = RT_CHILD_NONE;
    }
#endif
    if (ref[0] != RT_CHILD_NONE) cur = ref[0];
    else cur = PREF ? stack_pop_pref(sp, ovf, tos_pref) : stack_pop(sp, stk, ovf);
}

// ---- the voted trip: ONE source for the loop body of every per-lane traversal (traverse<> and stream_body; round 3) --------
// Every trip the wave VOTES: either every lane that holds an inner node takes one node step (one 64-B fetch; wide tree: four
// quantised slab tests, binary tree: two), or every lane that holds a leaf tests ONE record of it -- whichever serves more
// lanes per instruction issued (n_node * VOTE_TRI >= n_leaf * VOTE_NODE picks the node step).  All loop control is
// wave-uniform.  Rounds 1 and 2 kept copies of this body in traverse<>, in k_stream and in a third kernel; a shared FUNCTION
// was tried and cost the path kernels registers, so the shared source is a macro: expanded in place, it
// compiles to what the copies compiled to.  Parameters: the tree (NODES: 64-B nodes of the kind WIDE_ says, RECS, MESHES),
// FLAGS, compile-time WIDE_ / COUNT_ / UNIFORM_ (UNIFORM_: when every lane that takes the node step holds the SAME node --
// 22-79 % of the node trips, the top of the tree walked in lockstep -- ONE scalar load brings it into SGPRs for the whole
// wave instead of four vector loads per lane; only k_stream has the SGPRs to spare), the lane's ray (O, D, INV), its closest
// hit so far (T, HU, HV, SEQ, REC), its place in the tree (CUR, SP, STK, OVF), ANY_ (any accepted hit ends the ray), the
// wave's lane count N_ALL and ended-lane count N_DONE (read; the caller recounts after the trip), the vote weights, UNI_MIN,
// the counters C and the lane id LANE.  The wave's priority follows the trip (RT_PRIO_*, above wide_decide).
#define RT_VOTED_TRIP(NODES, RECS, MESHES, FLAGS, WIDE_, COUNT_, UNIFORM_, PREF_, O, D, INV, T, HU, HV, SEQ, REC, CUR, SP, STK, OVF, ANY_, N_ALL, N_DONE, VOTE_NODE, VOTE_TRI, UNI_MIN, C, LANE) \
    { \
        RT_SETPRIO(RT_PRIO_ISSUE); \
        const unsigned long long m_node_ = __ballot((CUR) >= 0); \
        const uint32_t n_node_ = (uint32_t)__builtin_popcountll(m_node_); \
        const uint32_t n_leaf_ = (N_ALL) - (N_DONE) - n_node_; \
        if (n_node_ * (VOTE_TRI) >= n_leaf_ * (VOTE_NODE)) { \
            bool uniform_ = false; \
            int u0_ = 0; \
            if (UNIFORM_) { \
                u0_ = __builtin_amdgcn_readlane((CUR), __builtin_ctzll(m_node_)); \
                uniform_ = n_node_ >= (UNI_MIN) && __ballot((CUR) >= 0 && (CUR) != u0_) == 0ull; \
            } \
            if (UNIFORM_ && uniform_) { \
                typedef __attribute__((address_space(4))) const float cfloat_; \
                cfloat_ *sp_ = (cfloat_ *)(const void *)((NODES) + u0_); \
                q4 q0 = {sp_[0], sp_[1], sp_[2], sp_[3]}, q1 = {sp_[4], sp_[5], sp_[6], sp_[7]}; \
                q4 q2 = {sp_[8], sp_[9], sp_[10], sp_[11]}, q3 = {sp_[12], sp_[13], sp_[14], sp_[15]}; \
                const int tp_ = (PREF_) ? stack_peek_lds((SP), (STK)) : 0; \
                RT_SETPRIO(RT_PRIO_WAIT); \
                if ((CUR) >= 0) { \
                    if (COUNT_) { (C).nodes++; if ((LANE) == (uint32_t)__builtin_ctzll(m_node_)) { (C).wnodes++; (C).walive += n_node_ + n_leaf_; } } \
                    if (WIDE_) wide_decide<PREF_>(q0, q1, q2, q3, (O), (INV), (T), (CUR), (SP), (STK), (OVF), tp_); \
                    else node_decide<PREF_>(q0, q1, q2, q3, (O), (INV), (T), (CUR), (SP), (STK), (OVF), tp_); \
                } \
            } else if ((CUR) >= 0) { \
                const float4 *p_ = reinterpret_cast<const float4 *>((NODES) + (CUR)); \
                float4 q0 = p_[0], q1 = p_[1], q2 = p_[2], q3 = p_[3]; \
                const int tp_ = (PREF_) ? stack_peek_lds((SP), (STK)) : 0; \
                RT_SETPRIO(RT_PRIO_WAIT); \
                if (COUNT_) { (C).nodes++; if ((LANE) == (uint32_t)__builtin_ctzll(m_node_)) { (C).wnodes++; (C).walive += n_node_ + n_leaf_; } } \
                if (WIDE_) wide_decide<PREF_>(q0, q1, q2, q3, (O), (INV), (T), (CUR), (SP), (STK), (OVF), tp_); \
                else node_decide<PREF_>(q0, q1, q2, q3, (O), (INV), (T), (CUR), (SP), (STK), (OVF), tp_); \
            } \
        } else { \
            if ((CUR) < 0 && (CUR) != RT_DONE) { \
                if ((CUR) == RT_CHILD_NONE) (CUR) = stack_pop((SP), (STK), (OVF)); \
                else { \
                    const uint32_t code_ = (uint32_t)(~(CUR)); \
                    const uint32_t first_ = code_ >> 3, left_ = code_ & 7u; \
                    const float4 *r_ = reinterpret_cast<const float4 *>((RECS) + first_); \
                    float4 r0 = r_[0], r1 = r_[1], r2 = r_[2]; \
                    const int tp_ = (PREF_) ? stack_peek_lds((SP), (STK)) : 0; \
                    RT_SETPRIO(RT_PRIO_WAIT); \
                    if (COUNT_) { (C).tris++; if ((LANE) == (uint32_t)__builtin_ctzll(__ballot(true))) (C).wtris++; } \
                    float dst_, u_, v_; \
                    bool ok_ = prim_test((O), (D), r0, r1, r2, dst_, u_, v_); \
                    RT_SETPRIO(RT_PRIO_PUSH); \
                    const uint32_t sq_ = __float_as_uint(r2.w); \
                    ok_ = ok_ && (dst_ < (T) || (dst_ == (T) && sq_ < (SEQ))); \
                    if (ok_ && !((FLAGS) & RT_FLAG_NO_MESH_CULL) && __float_as_uint(r2.z) != RT_REC_SPHERE) {   /* per-mesh AABB cull of wgsl:155 */ \
                        const float4 *mp_ = reinterpret_cast<const float4 *>((MESHES) + __float_as_uint(r2.z)); \
                        float4 m1 = mp_[1], m2 = mp_[2]; \
                        ok_ = ray_aabb_ref((O), exact_inv<WIDE_>((INV), (D)), F3(m1.x, m1.y, m1.z), F3(m2.x, m2.y, m2.z)); \
                        if (COUNT_) (C).meshes++; \
                    } \
                    if (ok_) { (T) = dst_; (SEQ) = sq_; (REC) = first_; (HU) = u_; (HV) = v_; }   /* tri / mesh: read back from the record when the ray ends */ \
                    if (ok_ && (ANY_)) (CUR) = RT_DONE;                       /* shadow ray: any hit answers it */ \
                    else if (left_ == 0u) (CUR) = (PREF_) ? stack_pop_pref((SP), (OVF), tp_) : stack_pop((SP), (STK), (OVF)); \
                    else (CUR) = ~(int)(((first_ + 1u) << 3) | (left_ - 1u)); \
                } \
            } \
        } \
    }

// ---- BVH traversal (build-side design; reference src/bvh.rs:1 is empty) ----------------------
// Closest hit with the reference's tie rule restated order-independently: among equal t the
// lowest position in the (mesh, triangle) loop order wins (ray_tracing.wgsl:164).
// The per-mesh AABB cull of wgsl:155 is applied to every accepted candidate, so a triangle whose
// mesh box the ray misses is skipped exactly as the reference's outer loop skips it.
// ANY: stop at the first accepted hit (shadow rays: hit_triangles(ray).hit).
//
// Wave-level shape: every trip of the loop the wave VOTES -- either every lane that holds an inner
// node takes one node step, or every lane that holds a leaf tests ONE triangle -- so that each
// instruction issued serves as many lanes as possible; loop control is wave-uniform (ballots).
// (The first version was a plain per-lane loop; the compiler's structurizer turned it into a node
// loop that every stack pop left, and a triangle loop that ran with ~10 % of the lanes; written
// this way the same rays take 1.2-1.4x less time.  vote_node : vote_tri = 3 : 1 measured best: a
// triangle step as soon as a quarter of the unfinished rays wait at a leaf -- early hits shorten
// everything that follows.)
// All lanes that call this must call it together (it counts them with a ballot).
template <bool ANY, bool COUNT, bool WIDE, bool PREF = false>
__device__ __forceinline__ void traverse(const RtSceneDev &S, f3 o, f3 d, uint32_t flags, int *stk_generic, Hit &h, Cnt &c,
                                         uint32_t vote_node = 3u, uint32_t vote_tri = 1u) {
    h.hit = false; h.t = RT_INF; h.u = 0.0f; h.v = 0.0f; h.seq = 0xFFFFFFFFu; h.tri = 0xFFFFFFFFu; h.mesh = 0;
    if (S.n_nodes == 0) return;
    const f3 inv = ray_inv<WIDE>(d);
    const RtBvhNode *const nodes = WIDE ? reinterpret_cast<const RtBvhNode *>(S.wide) : S.nodes;   // both 64 B
    lds_int *stk = (lds_int *)stk_generic;
    int ovf_mem[RT_STACK_OVF];
    prv_int *ovf = (prv_int *)ovf_mem;
    const uint32_t n_all = (uint32_t)__builtin_popcountll(__ballot(true));
    int sp = 0;
    int cur = ray_has_nan(o, d) ? RT_DONE : 0;
    uint32_t n_done = (uint32_t)__builtin_popcountll(__ballot(cur == RT_DONE));
    if (n_done == n_all) return;
    uint32_t rec = 0xFFFFFFFFu;           // the accepted record: one register instead of hit flag + tri + mesh
    const uint32_t lane = threadIdx.x & 63u;
    do {
        RT_VOTED_TRIP(nodes, S.recs, S.meshes, flags, WIDE, COUNT, false, PREF, o, d, inv, h.t, h.u, h.v, h.seq, rec, cur, sp, stk, ovf, ANY,
                      n_all, n_done, vote_node, vote_tri, 65u, c, lane)
        n_done = (uint32_t)__builtin_popcountll(__ballot(cur == RT_DONE));
    } while (n_done != n_all);
    if (rec != 0xFFFFFFFFu) {
        float4 r2 = reinterpret_cast<const float4 *>(S.recs + rec)[2];
        h.hit = true; h.tri = __float_as_uint(r2.y); h.mesh = __float_as_uint(r2.z);
    }
}


// ---- wave-coherent (packet) BVH traversal: one node fetch per WAVE, through the scalar cache ------
// Primary rays of an 8x8 pixel tile, and their shadow rays toward +Y, walk nearly the same
// subtrees.  Here the wave walks the tree as one unit: the node index is wave-uniform, the 64-B
// node is read ONCE per wave (one coalesced request instead of a gather of 64 different lines),
// every lane tests its own ray against the node's two child boxes straight from SGPR operands, and the wave descends where any lane hits
// (ballot).  The traversal stack is wave-uniform too (one small LDS column per wave).
// Per-lane results are identical to traverse<>: the same triangle arithmetic, the same
// order-independent closest-hit rule (lowest t, then lowest seq), conservative box culls only.
// Wave-uniform record fetch: lane l reads dword (l & 15) of the record -- ONE coalesced 64-B
// vector-memory request per wave -- and the dwords are then broadcast to SGPRs with v_readlane.
// (Scalar-cache s_load was tried first: its miss path serialises and ran 1.4x slower than the
// per-lane kernel on a 7.2 M-triangle scene; the vector L1 keeps many misses in flight.)
__device__ __forceinline__ float wave_fetch16(const void *rec) {
    return reinterpret_cast<const float *>(rec)[threadIdx.x & 15u];
}
__device__ __forceinline__ float bcast(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
#define RT_BCAST4(v, i) q4{bcast(v, 4 * (i)), bcast(v, 4 * (i) + 1), bcast(v, 4 * (i) + 2), bcast(v, 4 * (i) + 3)}

template <bool ANY, bool COUNT>
__device__ __forceinline__ void traverse_packet(const RtSceneDev &S, f3 o, f3 d, bool alive, uint32_t flags,
                                                int *wstk, Hit &h, Cnt &c) {
    h.hit = false; h.t = RT_INF; h.u = 0.0f; h.v = 0.0f; h.seq = 0xFFFFFFFFu; h.tri = 0xFFFFFFFFu; h.mesh = 0;
    if (S.n_nodes == 0) return;
    alive = alive && !ray_has_nan(o, d);
    if (__ballot(alive) == 0ull) return;
    const f3 inv = F3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    const bool lane0 = (threadIdx.x & 63u) == 0u;
    int sp = 0;
    int cur = 0;
    uint32_t last_mesh = 0xFFFFFFFFu;
    bool last_ok = true;
    float limit = RT_INF;
    for (;;) {
        cur = __builtin_amdgcn_readfirstlane(cur);
        if (cur >= 0) {
            float nv = wave_fetch16(S.nodes + cur);
            q4 q0 = RT_BCAST4(nv, 0), q1 = RT_BCAST4(nv, 1), q2 = RT_BCAST4(nv, 2), q3 = RT_BCAST4(nv, 3);
            if (COUNT) c.nodes += lane0 ? 1u : 0u;
            float ax0 = (q0.x - o.x) * inv.x, ax1 = (q0.w - o.x) * inv.x;
            float ay0 = (q0.y - o.y) * inv.y, ay1 = (q1.x - o.y) * inv.y;
            float az0 = (q0.z - o.z) * inv.z, az1 = (q1.y - o.z) * inv.z;
            float bx0 = (q1.z - o.x) * inv.x, bx1 = (q2.y - o.x) * inv.x;
            float by0 = (q1.w - o.y) * inv.y, by1 = (q2.z - o.y) * inv.y;
            float bz0 = (q2.x - o.z) * inv.z, bz1 = (q2.w - o.z) * inv.z;
            float tmin0 = fmaxf(fmaxf(fminf(ax0, ax1), fminf(ay0, ay1)), fminf(az0, az1));
            float tmax0 = fminf(fminf(fmaxf(ax0, ax1), fmaxf(ay0, ay1)), fmaxf(az0, az1));
            float tmin1 = fmaxf(fmaxf(fminf(bx0, bx1), fminf(by0, by1)), fminf(bz0, bz1));
            float tmax1 = fminf(fminf(fmaxf(bx0, bx1), fmaxf(by0, by1)), fmaxf(bz0, bz1));
            tmax0 *= 1.0000005f; tmax1 *= 1.0000005f;
            bool h0 = alive && tmin0 <= tmax0 && tmax0 >= 0.0f && tmin0 <= limit;
            bool h1 = alive && tmin1 <= tmax1 && tmax1 >= 0.0f && tmin1 <= limit;
            unsigned long long m0 = __ballot(h0), m1 = __ballot(h1);
            int c0 = __float_as_int(q3.x), c1 = __float_as_int(q3.y);
            if (m0 && m1) {
                // near child by vote of the lanes that hit something
                unsigned long long w1 = __ballot(h1 && (!h0 || tmin1 < tmin0));
                bool swap = 2 * __builtin_popcountll(w1) > __builtin_popcountll(m0 | m1);
                int nearc = swap ? c1 : c0, farc = swap ? c0 : c1;
                if (lane0) wstk[sp] = farc;
                ++sp;
                cur = nearc;
                continue;
            }
            if (m0) { cur = c0; continue; }
            if (m1) { cur = c1; continue; }
        } else if (cur != RT_CHILD_NONE) {
            uint32_t code = (uint32_t)(~cur);
            uint32_t first = code >> 3, n = (code & 7u) + 1u;
            for (uint32_t k = 0; k < n; ++k) {
                // 48-B record: dwords 12..15 of the 64-B window belong to the next record (or the
                // 16-B tail pad the upload adds); they are fetched and ignored
                float rv = wave_fetch16(S.recs + first + k);
                q4 r0 = RT_BCAST4(rv, 0), r1 = RT_BCAST4(rv, 1), r2 = RT_BCAST4(rv, 2);
                if (COUNT) c.tris += lane0 ? 1u : 0u;
                float dst, u, v;
                bool ok = prim_test(o, d, r0, r1, r2, dst, u, v);
                uint32_t seq = __float_as_uint(r2.w);
                ok = ok && alive && (dst < h.t || (dst == h.t && seq < h.seq));
                if (__ballot(ok) == 0ull) continue;
                uint32_t mesh = __float_as_uint(r2.z);
                if (!(flags & RT_FLAG_NO_MESH_CULL) && mesh != RT_REC_SPHERE) {
                    if (mesh != last_mesh) {      // wave-uniform: the whole wave looks at one triangle
                        float mv = wave_fetch16(S.meshes + mesh);   // 48-B MeshInfo, same 64-B window rule
                        q4 m1v = RT_BCAST4(mv, 1), m2v = RT_BCAST4(mv, 2);
                        last_ok = ray_aabb_ref(o, inv, F3(m1v.x, m1v.y, m1v.z), F3(m2v.x, m2v.y, m2v.z));
                        last_mesh = mesh;
                        if (COUNT) c.meshes += lane0 ? 1u : 0u;
                    }
                    ok = ok && last_ok;
                }
                if (ok) {
                    h.hit = true; h.t = dst; h.u = u; h.v = v; h.seq = seq;
                    h.tri = __float_as_uint(r2.y); h.mesh = mesh;
                    limit = dst * 1.00001f;
                    if (ANY) alive = false;       // this lane's shadow ray is answered
                }
            }
            if (ANY && __ballot(alive) == 0ull) return;
        }
        if (sp == 0) break;
        --sp;
        cur = wstk[sp];
    }
}

// ---- A5 reference-shaped closest hit (ray_tracing.wgsl:150-170): meshes, AABB cull, triangles ----
template <bool ANY, bool COUNT>
__device__ __forceinline__ void brute_force(const RtSceneDev &S, f3 o, f3 d, uint32_t flags, Hit &h, Cnt &c) {
    h.hit = false; h.t = 0.0f; h.u = 0.0f; h.v = 0.0f; h.seq = 0xFFFFFFFFu; h.tri = 0xFFFFFFFFu; h.mesh = 0;
    const f3 inv = F3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    for (uint32_t i = 0; i < S.n_meshes; ++i) {
        const float4 *mp = reinterpret_cast<const float4 *>(S.meshes + i);
        float4 m0 = mp[0], m1 = mp[1], m2 = mp[2];
        if (COUNT) c.meshes++;
        if (!(flags & RT_FLAG_NO_MESH_CULL) &&
            !ray_aabb_ref(o, inv, F3(m1.x, m1.y, m1.z), F3(m2.x, m2.y, m2.z)))
            continue;
        uint32_t first = __float_as_uint(m0.x), count = __float_as_uint(m0.y);
        for (uint32_t j = first; j < first + count; ++j) {
            rt_triangle T = S.tris[j];
            const float4 *pa = reinterpret_cast<const float4 *>(S.verts + T.indices[0]);
            const float4 *pb = reinterpret_cast<const float4 *>(S.verts + T.indices[1]);
            const float4 *pc = reinterpret_cast<const float4 *>(S.verts + T.indices[2]);
            float4 a = pa[0], b = pb[0], cc = pc[0];
            if (COUNT) c.tris++;
            f3 A = F3(a.x, a.y, a.z);
            float dst, u, v;
            bool ok = ray_tri(o, d, A, sub(F3(b.x, b.y, b.z), A), sub(F3(cc.x, cc.y, cc.z), A), dst, u, v);
            if (!ok) continue;
            if (!h.hit || h.t > dst) {   // strict: the earlier triangle keeps a tie (wgsl:164)
                h.hit = true; h.t = dst; h.u = u; h.v = v; h.tri = j; h.mesh = i;
                if (ANY) return;
            }
        }
    }
    for (uint32_t i = 0; i < S.n_spheres; ++i) {   // extension: spheres come after every mesh in loop order
        float4 sp = *reinterpret_cast<const float4 *>(S.spheres + i);
        if (COUNT) c.tris++;
        float dst;
        if (!ray_sphere(o, d, F3(sp.x, sp.y, sp.z), sp.w, dst)) continue;
        if (!h.hit || h.t > dst) {
            h.hit = true; h.t = dst; h.u = 0.0f; h.v = 0.0f; h.tri = S.n_tris + i; h.mesh = RT_REC_SPHERE;
            if (ANY) return;
        }
    }
}

// PREF: pops take the LDS stack's top entry from a register filled at the start of the trip (stack_pop_pref).  On for the
// bounce segments of the path tracer, whose incoherent rays wait for memory with few lanes alive (10 M triangles: + 3.6 %); off for
// coherent rays, where the extra LDS read per trip costs more than the shorter pop saves (k_stream + 0.5 ... + 1.5 %;
// profiles/r3/ab_tos_prefetch.txt).
template <bool ANY, bool COUNT, bool PREF = false>
__device__ __forceinline__ void trace(const RtSceneDev &S, f3 o, f3 d, uint32_t flags, int *stk, Hit &h, Cnt &c) {
    if (flags & RT_FLAG_BRUTE_FORCE) brute_force<ANY, COUNT>(S, o, d, flags, h, c);
    else if (S.n_wide != 0u && !(flags & RT_FLAG_BINARY_TREE)) traverse<ANY, COUNT, true, PREF>(S, o, d, flags, stk, h, c);   // kernel-uniform choice
    else traverse<ANY, COUNT, false, PREF>(S, o, d, flags, stk, h, c);
}

// point and interpolated vertex normal of an accepted hit: ray_tracing.wgsl:137-138
__device__ __forceinline__ void finish_hit(const RtSceneDev &S, f3 o, f3 d, const Hit &h, f3 &point, f3 &normal,
                                           uint32_t &material) {
    if (h.mesh == RT_REC_SPHERE) {      // extension: normal = normalize(point - centre)
        uint32_t i = h.tri - S.n_tris;
        float4 sp = *reinterpret_cast<const float4 *>(S.spheres + i);
        point = add(o, scl(d, h.t));
        normal = normalize(sub(point, F3(sp.x, sp.y, sp.z)));
        material = S.sphere_mats[i];
        return;
    }
    rt_triangle T = S.tris[h.tri];
    const float4 *pa = reinterpret_cast<const float4 *>(S.verts + T.indices[0]);
    const float4 *pb = reinterpret_cast<const float4 *>(S.verts + T.indices[1]);
    const float4 *pc = reinterpret_cast<const float4 *>(S.verts + T.indices[2]);
    float4 na = pa[1], nb = pb[1], nc = pc[1];
    float w = 1.0f - h.u - h.v;
    point = add(o, scl(d, h.t));
    f3 n = add(add(scl(F3(na.x, na.y, na.z), w), scl(F3(nb.x, nb.y, nb.z), h.u)), scl(F3(nc.x, nc.y, nc.z), h.v));
    normal = normalize(n);
    material = S.meshes[h.mesh].material;
}

// ---- pixel mapping: shard tiles -> 16x16 blocks -> 8x8 waves; XCD-aware block order -----------
// Blocks b and b+8 share an XCD (and its 4 MiB L2) under round-robin dispatch.  All 16 blocks of
// one 64x64 shard tile are given the same b % 8, so the rays that walk the same BVH subtrees hit
// the same L2, while consecutive tiles rotate over the 8 XCDs so that cheap regions (sky) and
// expensive ones (dense geometry) are spread evenly over the chip.  (Giving every XCD one
// contiguous screen band was measured first: the sky XCDs idled, 2.5 waves/SIMD on average.)
// Speed only, never correctness: the mapping is a bijection whatever the placement.
__device__ __forceinline__ bool pixel_of_thread(const RtViewDev &V, const RtShardDev &sh, uint32_t nb,
                                                uint32_t &px, uint32_t &py, uint32_t b = blockIdx.x) {
    (void)nb;
    uint32_t sub = sh.tile / RT_WORK_TILE;
    uint32_t per_tile = sub * sub;
    uint32_t xcd = b & 7u, j = b >> 3;
    uint32_t k = (j / per_tile) * 8u + xcd;      // owned-tile ordinal of this block
    uint32_t s = j % per_tile;                   // 16x16 sub-block inside the tile
    px = 0; py = 0;
    if (k >= sh.n_owned) return false;
    uint32_t g = sh.rank + k * (sh.count ? sh.count : 1u);
    uint32_t gx = g % sh.tiles_x, gy = g / sh.tiles_x;
    uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    uint32_t x = gx * sh.tile + (s % sub) * RT_WORK_TILE + (wave % (RT_WORK_TILE / 8u)) * 8u + (lane & 7u);
    uint32_t y = gy * sh.tile + (s / sub) * RT_WORK_TILE + (wave / (RT_WORK_TILE / 8u)) * 8u + (lane >> 3);
    px = V.vx + x; py = V.vy + y;
    return x < V.vw && y < V.vh;
}

__device__ __forceinline__ void wave_count(unsigned long long *ctr, bool pred) {
    unsigned long long m = __ballot(pred);
    if (m && (threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(m))
        atomicAdd(ctr, (unsigned long long)__builtin_popcountll(m));
}
__device__ __forceinline__ void wave_sum(unsigned long long *ctr, uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    if ((threadIdx.x & 63u) == 0 && v) atomicAdd(ctr, (unsigned long long)v);
}

// ---- the live path: ray_tracing.wgsl:173-211 (+ shadow ray) ------------------
__device__ __forceinline__ float4 shade_hit(const RtSceneDev &S, f3 normal, uint32_t material) {
    if (material >= 100u) {                               // debug grey (wgsl:205-207)
        float g = 0.5f + (float)(material - 100u) / 6.0f;
        return make_float4(g, g, g, g);
    }
    float ndl = dot(normal, F3(0.0f, 1.0f, 0.0f));
    float4 col = material < S.n_mats ? *reinterpret_cast<const float4 *>(S.mats + material)
                                     : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    return make_float4(col.x * ndl, col.y * ndl, col.z * ndl, col.w * ndl);   // alpha scaled too (wgsl:208)
}

// PACKET = wave-coherent traversal (traverse_packet); otherwise per-lane traversal with the LDS
// stack (or the reference-shaped loop under RT_FLAG_BRUTE_FORCE).
template <bool COUNT, bool PACKET>
__global__ __launch_bounds__(RT_BLOCK, RT_MIN_WAVES) void k_primary(RtSceneDev S, RtViewDev V, RtShardDev sh, uint32_t nb,
                                                      uint32_t flags, float4 *__restrict__ rgba,
                                                      uint32_t *__restrict__ prim, float *__restrict__ tout,
                                                      unsigned long long *__restrict__ counters) {
    __shared__ int s_stack[PACKET ? (RT_BLOCK / 64) * (RT_MAX_DEPTH + 2) : RT_STACK_LDS * RT_BLOCK];
    int *stk = PACKET ? s_stack + (threadIdx.x >> 6) * (RT_MAX_DEPTH + 2) : s_stack + threadIdx.x;
    uint32_t px, py;
    bool active = pixel_of_thread(V, sh, nb, px, py);
    Cnt c; c.nodes = 0; c.tris = 0; c.meshes = 0; c.wnodes = 0; c.wtris = 0; c.walive = 0;
    f3 o = F3(0.0f, 0.0f, 0.0f), d = F3(0.0f, 0.0f, 1.0f);
    if (active) ray_gen(V, px, py, o, d);
    Hit h;
    if (PACKET) traverse_packet<false, COUNT>(S, o, d, active, flags, stk, h, c);
    else if (active) trace<false, COUNT>(S, o, d, flags, stk, h, c);
    else h.hit = false;
    float4 out = make_float4(0.0f, 0.0f, 0.0f, 0.0f);        // miss: vec4(0) (wgsl:210)
    bool hit = active && h.hit;
    f3 point = F3(0.0f, 0.0f, 0.0f), normal = F3(0.0f, 1.0f, 0.0f);
    if (hit) {
        uint32_t material;
        finish_hit(S, o, d, h, point, normal, material);
        out = shade_hit(S, normal, material);
    }
    bool did_shadow = hit && (flags & RT_FLAG_SHADOW);
    if (flags & RT_FLAG_SHADOW) {
        f3 so = add(point, scl(normal, 1e-3f));
        Hit sh_hit;
        if (PACKET) traverse_packet<true, COUNT>(S, so, F3(0.0f, 1.0f, 0.0f), did_shadow, flags, stk, sh_hit, c);
        else if (did_shadow) trace<true, COUNT>(S, so, F3(0.0f, 1.0f, 0.0f), flags, stk, sh_hit, c);
        else sh_hit.hit = false;
        if (did_shadow && sh_hit.hit) out = make_float4(out.x * 0.25f, out.y * 0.25f, out.z * 0.25f, out.w * 0.25f);
    }
    if (active) {
        size_t idx = (size_t)py * V.width + px;
        rgba[idx] = out;
        prim[idx] = hit ? h.tri : 0xFFFFFFFFu;
        tout[idx] = hit ? h.t : 0.0f;
    }
    wave_count(counters + RT_CNT_PRIMARY, active);
    wave_count(counters + RT_CNT_SHADOW, did_shadow);
    if (COUNT) {
        wave_sum(counters + RT_CNT_NODES, c.nodes);
        wave_sum(counters + RT_CNT_TRIS, c.tris);
        wave_sum(counters + RT_CNT_MESH, c.meshes);
        wave_sum(counters + RT_CNT_WNODE, c.wnodes);
        wave_sum(counters + RT_CNT_WTRI, c.wtris);
        wave_sum(counters + RT_CNT_WALIVE, c.walive);
    }
}


// ---- k_stream: persistent waves, per-XCD work queues, phase voting, batched hand-over ---------------
// What RT_FLAG_COUNTERS shows for k_primary on the 7.2 M-triangle bench scene: a wave lives as long as
// its LONGEST ray, so 46 % of its node-loop trips run with <= 8 of the 64 rays still unfinished (58 %
// with <= 16), its triangle loop runs with ~10 % of the lanes, and every lane waits at the end of the
// primary phase for the wave's last primary ray before any shadow ray starts.  Here
//   * waves are persistent; a lane is a little state machine (idle / primary ray / shadow ray), a lane
//     whose primary ray ends starts its shadow ray without waiting for the wave;
//   * every trip of the main loop the wave VOTES: either every lane that holds an inner node takes one
//     node step, or every lane that holds a leaf tests ONE triangle -- whichever serves more lanes per
//     instruction issued (vote weights: RtTune).  All loop control is wave-uniform (ballots), nothing is left to the
//     compiler's structurizer (it re-shapes k_primary's loop from one build to the next);
//   * ended rays are handed over in batches (shade + write + start the shadow ray + take new pixels),
//     so the divergent finishing code runs for many lanes at once;
//   * work comes from EIGHT queues, one per XCD (read from HW_REG_XCC_ID): queue x holds the 64x64
//     shard tiles k with k % 8 == x as 16x16-pixel regions; a wave owns a region until its 256 pixels
//     are handed out (one atomic per 256 pixels; one queue head would saturate at ~88 dequeues/us:
//     MI355X guide, row 'dequeue').  A wave whose home queue is empty steals from the next ones.
//     refill_min = 64 hands a wave one whole 8x8 pixel block at a time (rays stay in lockstep, the
//     vector L1 serves most node fetches); smaller values refill single lanes (fewer idle lanes, but
//     desynchronised rays: measured 2.7x more L2 requests and slower on BVHs beyond the L2);
//     Measured and dropped: PARKING a wave's last <= 8 rays for a second launch that packs the stragglers of
//     all waves (fewer loop trips, but that launch runs few, long, incoherent rays: 1.47 vs 1.33 ms), and
//     handing a block's four waves the four 8x8 blocks of one 16x16 quad (coarser work units balance worse:
//     1.9 vs 1.33 ms at 1080p, where a wave only sees ~4 regions).
// Per-pixel results do not depend on any of this: same arithmetic, same closest-hit rule.
__device__ __forceinline__ uint32_t xcc_id() {
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(v));
    return v & 7u;
}
// Region r of queue x, pixel w of the region -> screen pixel.  A region is one 8x8 pixel block; a
// queue lists the blocks of its tiles so that four consecutive regions form a 16x16 quad (waves that
// dequeue one after the other work next to each other, as the blocks of k_primary's grid do; with
// 16x16 regions four times as much screen -- and BVH -- was in flight per XCD and the L2 missed more).
#define RT_REGION_PX 64u
__device__ __forceinline__ bool pixel_of_region(const RtViewDev &V, const RtShardDev &sh, uint32_t x, uint32_t r,
                                                uint32_t w, uint32_t &px, uint32_t &py) {
    uint32_t sub = sh.tile >> 4;                 // 16x16 quads per tile edge
    uint32_t rpt = sub * sub * 4u;               // regions per tile
    uint32_t k = (r / rpt) * 8u + x, rr = r % rpt;
    uint32_t g = sh.rank + k * (sh.count ? sh.count : 1u);
    uint32_t gx = g % sh.tiles_x, gy = g / sh.tiles_x;
    uint32_t quad = rr >> 2, b = rr & 3u;
    uint32_t xx = gx * sh.tile + (quad % sub) * 16u + (b & 1u) * 8u + (w & 7u);
    uint32_t yy = gy * sh.tile + (quad / sub) * 16u + (b >> 1) * 8u + (w >> 3);
    px = V.vx + xx; py = V.vy + yy;
    return xx < V.vw && yy < V.vh;
}

// (Round 2 also carried a block TAIL POOL here -- a wave down to its last <= 16 rays deposited them, stack and all, in an LDS
// pool of its block and took a fresh region; an idle wave took up to 64 pooled stragglers at once.  Frames and counted visits
// identical, VALU instructions -12 %, lanes active 0.47 -> 0.56, and the frame +3 % one at a time, +13 % with frames in flight
// (24.5 KB of LDS and 88 VGPRs: 5 waves per SIMD).  Measured, kept as an opt-in for a round, removed in round 3; the code is
// in the history at fa55bd4, the measurements in profiles/r2/ab_pool_*.txt and DESIGN.md section 4.1.)
struct RtStreamTune {
    uint32_t refill_min, finish_min, vote_node, vote_tri, uni_min;
};

template <bool COUNT, bool WIDE>
__device__ __forceinline__ void stream_body(
    const RtSceneDev &S_, const RtViewDev &V_, const RtShardDev &sh_, uint32_t flags, const RtStreamTune tn,
    float4 *__restrict__ rgba, uint32_t *__restrict__ prim, float *__restrict__ tout, unsigned long long *__restrict__ counters) {
    __shared__ int s_stack[RT_STACK_LDS * RT_BLOCK];
    // what only the hand-over code reads lives in LDS, not in ~60 SGPRs that the hot loop would spill
    __shared__ RtSceneDev S;
    __shared__ RtViewDev V;
    __shared__ RtShardDev sh;
    if (threadIdx.x == 0) { S = S_; V = V_; sh = sh_; }
    __syncthreads();
    lds_int *stk = (lds_int *)(s_stack + threadIdx.x);
    int ovf_mem[RT_STACK_OVF];
    prv_int *ovf = (prv_int *)ovf_mem;
    // the wide tree's nodes are 64 B as well: the fetch code is shared, only the decision differs
    const RtBvhNode *const nodes = WIDE ? reinterpret_cast<const RtBvhNode *>(S_.wide) : S_.nodes;
    const RtTriRecord *const recs = S_.recs;
    const rt_mesh_info *const meshes = S_.meshes;
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long below = (1ull << lane) - 1ull;
    const uint32_t home = xcc_id();
    unsigned long long *heads = counters + RT_CNT_HEADS;
    const uint32_t refill_min = tn.refill_min, finish_min = tn.finish_min, vote_node = tn.vote_node, vote_tri = tn.vote_tri, uni_min = tn.uni_min;
    // wave-uniform work-source state
    uint32_t reg_q = 0, reg_r = 0, reg_next = RT_REGION_PX;  // the wave's region and its next pixel
    uint32_t qoff = 0u;                                     // queues found empty so far (8 = all)
    Cnt c; c.nodes = 0; c.tris = 0; c.meshes = 0; c.wnodes = 0; c.wtris = 0; c.walive = 0;
    uint32_t n_primary = 0, n_shadow = 0;
    // per-lane ray
    f3 o = F3(0.0f, 0.0f, 0.0f), d = F3(0.0f, 0.0f, 1.0f), inv = F3(0.0f, 0.0f, 1.0f);
    float t = RT_INF, hu = 0.0f, hv = 0.0f;
    uint32_t seq = 0xFFFFFFFFu, rec = 0xFFFFFFFFu;
    int cur = RT_DONE, sp = 0;
    int state = 0;                 // 0 idle, 1 primary ray, 2 shadow ray
    uint32_t idx = 0;
#ifdef RT_TRIP_LIMIT
    uint32_t dbg_trips = 0;
#endif
#ifdef RT_DEBUG_TIMELINE
    const unsigned long long dbg_t0 = wall_clock64();
#endif
    for (;;) {
        // ================= service: runs when the hot loop below asks for it =================================
        // What the service code reads of the scene / view / shard descriptions it reads from LDS HERE, every time: without
        // this barrier the compiler loads all of it once at kernel start, into ~40 VGPRs that it then has to spill
        // (every lane of every wave storing 176 B at start-up: 80 MB of a frame's 203 MB of writes in round 1).
        __asm__ volatile("" ::: "memory");
        RT_SETPRIO(RT_PRIO_ISSUE);
        // ---- 1. finish ended rays (shade, write, start the shadow ray) -----------------------------------------
        if (cur == RT_DONE && state == 1) {
            float4 out = make_float4(0.0f, 0.0f, 0.0f, 0.0f);      // miss: vec4(0) (wgsl:210)
            uint32_t out_prim = 0xFFFFFFFFu; float out_t = 0.0f;
            state = 0;
            if (rec != 0xFFFFFFFFu) {
                float4 r2 = reinterpret_cast<const float4 *>(recs + rec)[2];
                Hit h;
                h.hit = true; h.t = t; h.u = hu; h.v = hv; h.seq = seq;
                h.tri = __float_as_uint(r2.y); h.mesh = __float_as_uint(r2.z);
                f3 point, normal; uint32_t material;
                finish_hit(S, o, d, h, point, normal, material);
                out = shade_hit(S, normal, material);
                out_prim = h.tri; out_t = t;
                if (flags & RT_FLAG_SHADOW) {
                    o = add(point, scl(normal, 1e-3f)); d = F3(0.0f, 1.0f, 0.0f);
                    inv = ray_inv<WIDE>(d);
                    t = RT_INF; seq = 0xFFFFFFFFu; rec = 0xFFFFFFFFu;
                    cur = ray_has_nan(o, d) ? RT_DONE : 0; sp = 0;
                    state = 2;
                    ++n_shadow;
                }
            }
            // the unshadowed value is stored now; an occluded shadow ray scales it in place later
            rgba[idx] = out; prim[idx] = out_prim; tout[idx] = out_t;
        } else if (cur == RT_DONE && state == 2) {
            if (rec != 0xFFFFFFFFu) {
                float4 out = rgba[idx];
                rgba[idx] = make_float4(out.x * 0.25f, out.y * 0.25f, out.z * 0.25f, out.w * 0.25f);
            }
            state = 0;
        }
        bool more = qoff < 8u || reg_next < RT_REGION_PX;
        // ---- 2. idle lanes take new work -------------------------------------------------------------------------
        unsigned long long idle = __ballot(state == 0);
        {
            // the wave's region; a used-up region is replaced from the XCD's queue (one atomic per 64 pixels), an empty queue
            // by the next XCD's
            const bool all_idle = idle == ~0ull;
            const uint32_t rpt = (sh.tile >> 4) * (sh.tile >> 4) * 4u;   // 8x8 regions per shard tile
            while (idle && ((uint32_t)__builtin_popcountll(idle) >= refill_min || all_idle)) {
                if (reg_next >= RT_REGION_PX) {               // wave-uniform
                    if (qoff >= 8u) break;
                    const uint32_t q = (home + qoff) & 7u;
                    const uint32_t n_tiles_q = sh.n_owned > q ? (sh.n_owned - q + 7u) / 8u : 0u;
                    const uint32_t len = n_tiles_q * rpt;
                    uint32_t r = len;
                    if (len != 0u) {
                        if (lane == 0u) r = (uint32_t)atomicAdd(heads + q * 16u, 1ull);
                        r = (uint32_t)__builtin_amdgcn_readfirstlane((int)r);
                    }
                    if (r >= len) {
                        // this queue is empty: ONE look at all eight heads (lanes 0..7), then on to the first queue
                        // that still has regions -- not seven more dependent atomics before a wave may leave
                        const uint32_t ql = (home + lane) & 7u;
                        const uint32_t len_l = (sh.n_owned > ql ? (sh.n_owned - ql + 7u) / 8u : 0u) * rpt;
                        bool avail = false;
                        if (lane < 8u && lane > qoff)
                            avail = __hip_atomic_load(heads + ql * 16u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned long long)len_l;
                        const unsigned long long ma = __ballot(avail);
                        qoff = ma ? (uint32_t)__builtin_ctzll(ma) : 8u;
                        continue;
                    }
                    reg_q = q; reg_r = r; reg_next = 0u;
                }
                if (state == 0) {
                    uint32_t w = reg_next + (uint32_t)__builtin_popcountll(idle & below);
                    uint32_t px, py;
                    if (w < RT_REGION_PX && pixel_of_region(V, sh, reg_q, reg_r, w, px, py)) {
                        ray_gen(V, px, py, o, d);
                        inv = ray_inv<WIDE>(d);
                        t = RT_INF; seq = 0xFFFFFFFFu; rec = 0xFFFFFFFFu;
                        cur = S.n_nodes && !ray_has_nan(o, d) ? 0 : RT_DONE; sp = 0;
                        idx = py * V.width + px;
                        state = 1;
                        ++n_primary;
                    }
                }
                reg_next += (uint32_t)__builtin_popcountll(idle);   // pixels outside the image are skipped
                idle = __ballot(state == 0);
            }
            more = qoff < 8u || reg_next < RT_REGION_PX;
            if (idle == ~0ull && !more) break;                // nothing in flight, no regions left: leave
        }
        if (__ballot(cur != RT_DONE) == 0ull) continue;   // only ended rays (e.g. an empty scene): service again
        // ================= hot loop: one voted step per trip, until the wave needs service ========================
        const bool any = state == 2;
        // idle lanes stay idle inside the hot loop, so ended-but-unfinished rays = done - idle: two ballots a trip
        const uint32_t n_idle = (uint32_t)__builtin_popcountll(__ballot(state == 0));
        uint32_t n_done = (uint32_t)__builtin_popcountll(__ballot(cur == RT_DONE)), n_pend, n_live;
        do {
            RT_VOTED_TRIP(nodes, recs, meshes, flags, WIDE, COUNT, true, false, o, d, inv, t, hu, hv, seq, rec, cur, sp, stk, ovf, any,
                          64u, n_done, vote_node, vote_tri, uni_min, c, lane)
            n_done = (uint32_t)__builtin_popcountll(__ballot(cur == RT_DONE));
            n_pend = n_done - n_idle;
            n_live = 64u - n_done;
#ifdef RT_TRIP_LIMIT      // debug builds only: a wave that loops too long reports a live lane's state and leaves
            if (++dbg_trips > RT_TRIP_LIMIT) {
                const unsigned long long live = __ballot(cur != RT_DONE);
                if (live && lane == (uint32_t)__builtin_ctzll(live)) {
                    counters[RT_CNT_BOUNCE] = ((unsigned long long)(uint32_t)cur << 32) | (uint32_t)sp;
                    counters[RT_CNT_MESH] = ((unsigned long long)__float_as_uint(t) << 32) | idx;
                }
                return;
            }
#endif
        } while (!(n_live == 0u || n_pend >= finish_min || (more && n_done >= refill_min)));
    }
#ifdef RT_DEBUG_TIMELINE     // debug builds only (tools/timeline.py): per-wave start / end times (100 MHz) over the hit-distance output
    if (lane == 0u) {
        uint32_t wid = blockIdx.x * (RT_BLOCK / 64u) + (threadIdx.x >> 6);
        reinterpret_cast<uint32_t *>(tout)[2u * wid] = (uint32_t)dbg_t0;
        reinterpret_cast<uint32_t *>(tout)[2u * wid + 1u] = (uint32_t)wall_clock64();
    }
#endif
    wave_sum(counters + RT_CNT_PRIMARY, n_primary);
    wave_sum(counters + RT_CNT_SHADOW, n_shadow);
    if (COUNT) {
        wave_sum(counters + RT_CNT_NODES, c.nodes);
        wave_sum(counters + RT_CNT_TRIS, c.tris);
        wave_sum(counters + RT_CNT_MESH, c.meshes);
        wave_sum(counters + RT_CNT_WNODE, c.wnodes);
        wave_sum(counters + RT_CNT_WTRI, c.wtris);
        wave_sum(counters + RT_CNT_WALIVE, c.walive);
    }
}

// The persistent kernel is built for 7 waves per SIMD (<= 72 VGPRs): same-box A/B against 8 waves / 64 VGPRs (more spills)
// and 6 waves / 80 VGPRs: default scene 0.233 / 0.255 / 0.240 ms, 1 M-triangle soup 0.596 / 0.606 / 0.591, 7.2 M-triangle bench
// scene 0.809 / 0.813 / 0.836.  (The path kernels: RT_MIN_WAVES above.)
#ifndef RT_STREAM_WAVES
#define RT_STREAM_WAVES RT_MIN_WAVES
#endif
template <bool COUNT, bool WIDE>
__global__ __launch_bounds__(RT_BLOCK, RT_STREAM_WAVES) void k_stream(
    RtSceneDev S_, RtViewDev V_, RtShardDev sh_, uint32_t flags, RtStreamTune tn, float4 *__restrict__ rgba, uint32_t *__restrict__ prim,
    float *__restrict__ tout, unsigned long long *__restrict__ counters) {
    stream_body<COUNT, WIDE>(S_, V_, sh_, flags, tn, rgba, prim, tout, counters);
}
// The same kernel built for 8 waves per SIMD (64 VGPRs, 10 spilled): launched with three or more frames in flight, where the
// chip never drains and one more wave per SIMD hides more of the gathers' latency than the spills cost -- bench scene, 4 in
// flight: 0.597 -> 0.582 ms per frame (1 M-triangle soup 0.394 -> 0.387); one frame at a time it is the same or 2 % slower,
// and 6 waves are 7 % slower in flight (profiles/r2/ab_waves_in_flight.txt).
template <bool COUNT, bool WIDE>
__global__ __launch_bounds__(RT_BLOCK, 8) void k_stream8(
    RtSceneDev S_, RtViewDev V_, RtShardDev sh_, uint32_t flags, RtStreamTune tn, float4 *__restrict__ rgba, uint32_t *__restrict__ prim,
    float *__restrict__ tout, unsigned long long *__restrict__ counters) {
    stream_body<COUNT, WIDE>(S_, V_, sh_, flags, tn, rgba, prim, tout, counters);
}

// ---- A2 RNG: ray_tracing.wgsl:38-74 (integer core; deterministic log/cos) ------------------------
__device__ __forceinline__ uint32_t next_random(uint32_t &state) {
    state = state * 747796405u + 2891336453u;
    uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
    return (word >> 22u) ^ word;
}
__device__ __forceinline__ float rt_rand(uint32_t &state) { return (float)next_random(state) / 4294967295.0f; }
__device__ __forceinline__ uint32_t pcg_hash(uint32_t x) {
    uint32_t st = x * 747796405u + 2891336453u;
    uint32_t word = ((st >> ((st >> 28u) + 4u)) ^ st) * 277803737u;
    return (word >> 22u) ^ word;
}
__device__ __forceinline__ uint32_t rt_seed(uint32_t pixel, uint32_t sample, uint32_t seed, uint32_t stream) {
    uint32_t h = pcg_hash(pixel);
    h = pcg_hash(h ^ (sample * 0x9E3779B9u + 0x7F4A7C15u));
    h = pcg_hash(h ^ (seed + 0x85EBCA6Bu));
    h = pcg_hash(h ^ (stream * 0xC2B2AE35u + 0x27D4EB2Fu));
    return h;
}
// log / cos from +,-,*,/ and integer bit moves only: CPU oracle and GPU agree bit for bit.
__device__ __forceinline__ float det_logf(float x) {
    if (x != x || x < 0.0f) return __uint_as_float(0x7FC00000u);
    if (x == 0.0f) return __uint_as_float(0xFF800000u);
    uint32_t ix = __float_as_uint(x);
    int k = 0;
    if (ix < 0x00800000u) { x = x * 8388608.0f; ix = __float_as_uint(x); k = -23; }
    if (ix >= 0x7F800000u) return x;
    ix += 0x3F800000u - 0x3F3504F3u;
    k += (int)(ix >> 23) - 127;
    ix = (ix & 0x007FFFFFu) + 0x3F3504F3u;
    float m = __uint_as_float(ix);
    float f = m - 1.0f;
    float s = f / (2.0f + f);
    float z = s * s;
    float w = z * z;
    float t1 = w * (0.40000972152f + w * 0.24279078841f);
    float t2 = z * (0.66666662693f + w * 0.28498786688f);
    float R = t2 + t1;
    float hfsq = 0.5f * f * f;
    float dk = (float)k;
    return dk * 6.9313812256e-01f + ((s * (hfsq + R) + dk * 9.0580006145e-06f) - hfsq + f);
}
__device__ __forceinline__ float det_cosf(float x) {
    if (x != x || x - x != 0.0f) return __uint_as_float(0x7FC00000u);
    float ax = x < 0.0f ? -x : x;
    float q = floorf(ax * 0.63661977237f + 0.5f);
    float r = ((ax - q * 1.5703125f) - q * 4.837512969970703125e-4f) - q * 7.54978995489188e-8f;
    int n = (int)q & 3;
    float r2 = r * r;
    float sp = r + r * r2 * (-1.6666654611e-1f + r2 * (8.3321608736e-3f + r2 * -1.9515295891e-4f));
    float cp = (1.0f - 0.5f * r2) + r2 * r2 * (4.166664568298827e-2f + r2 * (-1.388731625493765e-3f + r2 * 2.443315711809948e-5f));
    return n == 0 ? cp : (n == 1 ? -sp : (n == 2 ? -cp : sp));
}
__device__ __forceinline__ float rand_norm(uint32_t &state) {   // wgsl:57-61
    float theta = 2.0f * 3.1415926f * rt_rand(state);
    float rho = sqrtf(-2.0f * det_logf(rt_rand(state)));
    return rho * det_cosf(theta);
}
__device__ __forceinline__ f3 rand_hemi(uint32_t &state, f3 n) {  // wgsl:63-74
    float x = rand_norm(state);
    float y = rand_norm(state);
    float z = rand_norm(state);
    f3 dir = normalize(F3(x, y, z));
    if (dot(n, dir) < 0.0f) dir = F3(-dir.x, -dir.y, -dir.z);
    return dir;
}

// a queued path's pixel word: linear pixel index in the low bits, sample slot of the batch above them
#define RT_PATH_SLOT_SHIFT 26u
#define RT_PATH_PIXEL_MASK ((1u << RT_PATH_SLOT_SHIFT) - 1u)
// Wave-level compaction of surviving paths: one ballot, one atomic per wave.
__device__ __forceinline__ uint32_t queue_slot(unsigned long long *ctr, bool alive) {
    unsigned long long m = __ballot(alive);
    uint32_t lane = threadIdx.x & 63u;
    uint32_t base = 0;
    if (m) {
        int leader = __builtin_ctzll(m);
        if ((int)lane == leader) base = (uint32_t)atomicAdd(ctr, (unsigned long long)__builtin_popcountll(m));
        base = __shfl(base, leader);
    }
    return base + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull));
}

// One segment of one path (body of the loop at ray_tracing.wgsl:185-201): trace, then either add
// the sky term and end (miss) or scatter and keep going (hit).  Returns true if the path goes on.
// The traversal itself only needs the ray: colour, RNG state and pixel are fetched (or recomputed) by the callers AFTER
// trace<> returns, so that they do not occupy registers -- and get spilled -- inside the traversal loop.
__device__ __forceinline__ bool path_scatter(const RtSceneDev &S, const Hit &h, f3 &o, f3 &d, f3 &col, uint32_t &rng, float4 *acc) {
    if (h.hit) {
        f3 point, normal; uint32_t material;
        finish_hit(S, o, d, h, point, normal, material);
        f3 nd = rand_hemi(rng, normal);
        o = point; d = nd;
        if (material < S.n_mats) {
            float4 m = *reinterpret_cast<const float4 *>(S.mats + material);
            col = mul(col, F3(m.x, m.y, m.z));
        } else col = F3(0.0f, 0.0f, 0.0f);
        return true;
    }
    f3 l = mul(F3(0.5f, 0.71f, 0.86f), col);      // sky (wgsl:197)
    // light = vec3(0) + sky * col: the path's slot of the contribution plane holds (0, 0, 0, 1) since k_path_start (one
    // path per pixel and sample slot, no emission on the way: this is its only light), so the sum is formed from the
    // constant instead of an uncoalesced 16-B read -- the same addition, the same bits (0 + x keeps -0 apart from +0)
    *acc = make_float4(0.0f + l.x, 0.0f + l.y, 0.0f + l.z, 1.0f);
    return false;
}

// persist != 0: one chip-full of blocks strides over the frame's 16x16 pixel blocks; persist == 0
// (RT_FLAG_STATIC_GRID): one block per pixel block.  (Fed from k_stream's per-XCD queues instead, this kernel
// ran 10 % slower at 1080p than the static grid.)
// Samples are traced k x k at a time (k = 1 << kshift: 1, 4 or 16 samples of every pixel per launch): the launch runs on
// the frame magnified k times -- thread (sx, sy) of that grid is sample slot (sy % k) * k + sx % k of pixel (sx / k, sy / k)
// -- so a wave's 64 rays cover an eighth (a quarter ...) of the screen edge they would cover at one sample per pixel and
// walk the tree that much more in step, and a frame takes a quarter (a sixteenth) of the launches and of their tails.
// Slot s of a pixel adds its light to contrib[s * n_px + pixel] (written at most once per launch: one path per pixel and
// sample); k_accumulate_batch then adds the slots to the running sums in sample order -- the same additions in the same
// order as one launch per sample.
template <bool COUNT>
__global__ __launch_bounds__(RT_BLOCK, RT_MIN_WAVES) void k_path_start(RtSceneDev S, RtViewDev V, RtShardDev sh, uint32_t nb,
                                                         uint32_t flags, uint32_t sample, uint32_t seed,
                                                         uint32_t stream_id, uint32_t max_segments,
                                                         float4 *__restrict__ contrib, uint32_t n_px, uint32_t kshift, RtPathQueue qo,
                                                         unsigned long long *__restrict__ counters, int qo_ctr,
                                                         uint32_t persist) {
    __shared__ int s_stack[RT_STACK_LDS * RT_BLOCK];
    int *stk = s_stack + threadIdx.x;
    Cnt c; c.nodes = 0; c.tris = 0; c.meshes = 0; c.wnodes = 0; c.wtris = 0; c.walive = 0;
    uint32_t n_primary = 0;
    // grid-stride over the nb 16x16 pixel blocks of the frame: with persist the grid is one chip-full of
    // blocks (a multiple of 8, so a block keeps its XCD and the tile map of pixel_of_thread its meaning)
    // the magnified grid: viewport and shard tiles k times as large, the same tiles owned by the same ranks
    RtViewDev Vs = V;
    RtShardDev shs = sh;
    Vs.vx <<= kshift; Vs.vy <<= kshift; Vs.vw <<= kshift; Vs.vh <<= kshift; shs.tile <<= kshift;
    const uint32_t kmask = (1u << kshift) - 1u;
    for (uint32_t vb = blockIdx.x; vb < nb; vb += gridDim.x) {
        uint32_t px, py;
        bool active = pixel_of_thread(Vs, shs, nb, px, py, vb);
        const uint32_t slot_s = ((py & kmask) << kshift) | (px & kmask);
        px >>= kshift; py >>= kshift;
        bool alive = false;
        f3 o, d, col; uint32_t rng = 0, pixel = 0;
        if (active) {
            pixel = py * V.width + px;
            f3 d0;
            ray_gen(V, px, py, o, d0);
            uint32_t r0 = rt_seed(pixel, sample + slot_s, seed, stream_id);
            float jx = rt_rand(r0), jy = rt_rand(r0);
            d = add(d0, scl(F3(jx, jy, 0.0f), 0.0015f));   // wgsl:184, not renormalised
            contrib[(size_t)slot_s * n_px + pixel] = make_float4(0.0f, 0.0f, 0.0f, 1.0f);   // this pixel takes one more sample
            Hit h;
            trace<false, COUNT>(S, o, d, flags, stk, h, c);
            rng = rt_seed(pixel, sample + slot_s, seed, stream_id);       // recomputed, not kept across the traversal
            (void)rt_rand(rng); (void)rt_rand(rng);
            col = F3(1.0f, 1.0f, 1.0f);
            alive = path_scatter(S, h, o, d, col, rng, contrib + (size_t)slot_s * n_px + pixel) && max_segments > 1;
            ++n_primary;
        }
        uint32_t slot = queue_slot(counters + qo_ctr, alive);
        if (alive) {
            qo.ox[slot] = o.x; qo.oy[slot] = o.y; qo.oz[slot] = o.z;
            qo.dx[slot] = d.x; qo.dy[slot] = d.y; qo.dz[slot] = d.z;
            qo.cr[slot] = col.x; qo.cg[slot] = col.y; qo.cb[slot] = col.z;
            qo.rng[slot] = rng; qo.pixel[slot] = pixel | (slot_s << RT_PATH_SLOT_SHIFT);
        }
    }
    wave_sum(counters + RT_CNT_PRIMARY, n_primary);
    if (COUNT) {
        wave_sum(counters + RT_CNT_NODES, c.nodes);
        wave_sum(counters + RT_CNT_TRIS, c.tris);
        wave_sum(counters + RT_CNT_MESH, c.meshes);
        wave_sum(counters + RT_CNT_WNODE, c.wnodes);
        wave_sum(counters + RT_CNT_WTRI, c.wtris);
        wave_sum(counters + RT_CNT_WALIVE, c.walive);
    }
}

// persist != 0: one chip-full of blocks strides over the queue in 256-path chunks: later bounces hold few
// paths, and a grid sized for the frame would spend its time dispatching empty blocks.  (Taking 64 paths at
// a time from one work counter instead was 11 % slower on the 10 M-triangle scene.  A k_stream-style kernel
// that refills single lanes from the queue as their segments end -- bounce rays are incoherent, a wave's VALU
// instructions run with 36 % of the lanes -- was 1.1-1.4x SLOWER on the binary tree and 1.2-1.8x slower on the
// wide tree (refill thresholds 16-64, 256-path chunks per wave): lanes that start together fetch the same
// top-of-tree nodes in the same instruction, which the L1 front end serves as one line; desynchronised lanes
// do not.  DESIGN.md section 4.)
#ifndef RT_BOUNCE_PREF
#define RT_BOUNCE_PREF true
#endif
template <bool COUNT>
__global__ __launch_bounds__(RT_BLOCK, RT_MIN_WAVES) void k_path_bounce(RtSceneDev S, uint32_t flags, int is_last,
                                                          float4 *__restrict__ contrib, uint32_t n_px, RtPathQueue qi, RtPathQueue qo,
                                                          unsigned long long *__restrict__ counters, int qi_ctr,
                                                          int qo_ctr, uint32_t persist) {
    __shared__ int s_stack[RT_STACK_LDS * RT_BLOCK];
    int *stk = s_stack + threadIdx.x;
    const uint32_t n = (uint32_t)counters[qi_ctr];
    Cnt c; c.nodes = 0; c.tris = 0; c.meshes = 0; c.wnodes = 0; c.wtris = 0; c.walive = 0;
    uint32_t n_bounce = 0;
    (void)persist;
    for (uint32_t vb = blockIdx.x; vb * RT_BLOCK < n; vb += gridDim.x) {   // grid-stride over 256-path chunks
        const uint32_t i = vb * RT_BLOCK + threadIdx.x;
        bool active = i < n;
        bool alive = false;
        f3 o, d, col; uint32_t rng = 0, pixel = 0;
        if (active) {
            o = F3(qi.ox[i], qi.oy[i], qi.oz[i]);
            d = F3(qi.dx[i], qi.dy[i], qi.dz[i]);
            Hit h;
            trace<false, COUNT, RT_BOUNCE_PREF>(S, o, d, flags, stk, h, c);
            col = F3(qi.cr[i], qi.cg[i], qi.cb[i]);              // fetched after the traversal (registers)
            rng = qi.rng[i]; pixel = qi.pixel[i];                // pixel | sample slot << RT_PATH_SLOT_SHIFT
            alive = path_scatter(S, h, o, d, col, rng, contrib + (size_t)(pixel >> RT_PATH_SLOT_SHIFT) * n_px + (pixel & RT_PATH_PIXEL_MASK)) && !is_last;
            ++n_bounce;
        }
        uint32_t slot = queue_slot(counters + qo_ctr, alive);
        if (alive) {
            qo.ox[slot] = o.x; qo.oy[slot] = o.y; qo.oz[slot] = o.z;
            qo.dx[slot] = d.x; qo.dy[slot] = d.y; qo.dz[slot] = d.z;
            qo.cr[slot] = col.x; qo.cg[slot] = col.y; qo.cb[slot] = col.z;
            qo.rng[slot] = rng; qo.pixel[slot] = pixel;
        }
    }
    wave_sum(counters + RT_CNT_BOUNCE, n_bounce);
    if (COUNT) {
        wave_sum(counters + RT_CNT_NODES, c.nodes);
        wave_sum(counters + RT_CNT_TRIS, c.tris);
        wave_sum(counters + RT_CNT_MESH, c.meshes);
        wave_sum(counters + RT_CNT_WNODE, c.wnodes);
        wave_sum(counters + RT_CNT_WTRI, c.wtris);
        wave_sum(counters + RT_CNT_WALIVE, c.walive);
    }
}

// (Round 2 measured three re-shapings of the bounce segments and kept none as the default; round 3 removed their code, which is
// in the history at fa55bd4, measurements in profiles/r2 and DESIGN.md section 4.1: the queue regrouped by ray direction inside
// the kernel (k_path_bounce_sorted: -2 ... -3 % wave trips, +2 ... +5 % time), and the segment split into a lane-refilled
// traversal kernel + a scatter kernel (k_path_trace / k_path_shade: -38 % wave trips, +20 ... +30 % time).)

// ---- f1 refit on the GPU: vertices moved, topology unchanged -----------------------------
// One launch per tree level, deepest first (children are finished before their parent reads them; a kernel
// boundary on one stream is the only ordering needed).  A thread owns one node of the level: for a LEAF child
// it re-bakes the child's records from the new vertices (a, ab, ac: the values ray_tracing.wgsl:122-123
// computes) and boxes them with the builder's padding rule; for an INNER child it takes the union of that
// child's two boxes.  Same float operations as the host refit (bvh_build.cpp), so the boxes are bit-identical.
__device__ __forceinline__ void box_of_child(int32_t code, RtBvhNode *nodes, RtTriRecord *recs, const rt_triangle *tris,
                                             const rt_vertex *verts, float lo[3], float hi[3]) {
    for (int k = 0; k < 3; ++k) { lo[k] = RT_INF; hi[k] = -RT_INF; }
    if (code >= 0) {
        const RtBvhNode &C = nodes[code];
        if (C.child0 != RT_CHILD_NONE) for (int k = 0; k < 3; ++k) { lo[k] = fminf(lo[k], C.lo0[k]); hi[k] = fmaxf(hi[k], C.hi0[k]); }
        if (C.child1 != RT_CHILD_NONE) for (int k = 0; k < 3; ++k) { lo[k] = fminf(lo[k], C.lo1[k]); hi[k] = fmaxf(hi[k], C.hi1[k]); }
        return;
    }
    uint32_t v = (uint32_t)(~code), first = v >> 3, cnt = (v & 7u) + 1u;
    for (uint32_t i = first; i < first + cnt; ++i) {
        RtTriRecord &R = recs[i];
        float blo[3], bhi[3], amax = 0.0f;
        if (R.mesh == RT_REC_SPHERE) {           // spheres do not move in a vertex update
            float r = fabsf(R.abx);
            for (int k = 0; k < 3; ++k) amax = fmaxf(amax, fmaxf(fabsf(R.a[k] - r), fabsf(R.a[k] + r)));
            float pad = 2e-5f * amax + 1e-30f;
            for (int k = 0; k < 3; ++k) { blo[k] = R.a[k] - r - pad; bhi[k] = R.a[k] + r + pad; }
        } else {
            rt_triangle T = tris[R.tri];
            const float *a = verts[T.indices[0]].pos, *b = verts[T.indices[1]].pos, *c = verts[T.indices[2]].pos;
            float A[3] = {a[0], a[1], a[2]}, B[3] = {b[0], b[1], b[2]}, C[3] = {c[0], c[1], c[2]};
            for (int k = 0; k < 3; ++k) {
                blo[k] = fminf(A[k], fminf(B[k], C[k]));
                bhi[k] = fmaxf(A[k], fmaxf(B[k], C[k]));
                amax = fmaxf(amax, fmaxf(fabsf(blo[k]), fabsf(bhi[k])));
            }
            float pad = 1e-5f * amax + 1e-30f;    // same padding rule as the build
            for (int k = 0; k < 3; ++k) { blo[k] -= pad; bhi[k] += pad; }
            R.a[0] = A[0]; R.a[1] = A[1]; R.a[2] = A[2];
            R.abx = B[0] - A[0]; R.aby = B[1] - A[1]; R.abz = B[2] - A[2];
            R.acx = C[0] - A[0]; R.acy = C[1] - A[1]; R.acz = C[2] - A[2];
        }
        for (int k = 0; k < 3; ++k) { lo[k] = fminf(lo[k], blo[k]); hi[k] = fmaxf(hi[k], bhi[k]); }
    }
}
__global__ void k_refit_level(RtBvhNode *nodes, RtTriRecord *recs, const rt_triangle *__restrict__ tris,
                              const rt_vertex *__restrict__ verts, const uint32_t *__restrict__ level_nodes, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    RtBvhNode &N = nodes[level_nodes[i]];
    float lo[3], hi[3];
    if (N.child0 != RT_CHILD_NONE) {
        box_of_child(N.child0, nodes, recs, tris, verts, lo, hi);
        for (int k = 0; k < 3; ++k) { N.lo0[k] = lo[k]; N.hi0[k] = hi[k]; }
    }
    if (N.child1 != RT_CHILD_NONE) {
        box_of_child(N.child1, nodes, recs, tris, verts, lo, hi);
        for (int k = 0; k < 3; ++k) { N.lo1[k] = lo[k]; N.hi1[k] = hi[k]; }
    }
}
// Dirty-range refit (rt_update_vertex_range): the same boxes, computed only where something below moved.
__global__ void k_refit_level_dirty(RtBvhNode *nodes, RtTriRecord *recs, const rt_triangle *__restrict__ tris,
                                    const rt_vertex *__restrict__ verts, const uint32_t *__restrict__ level_nodes, uint32_t n,
                                    uint32_t mesh_lo, uint32_t mesh_hi, uint8_t *dirty) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t ni = level_nodes[i];
    RtBvhNode &N = nodes[ni];
    bool any = false;
    float lo[3], hi[3];
    for (int side = 0; side < 2; ++side) {
        const int32_t code = side ? N.child1 : N.child0;
        if (code == RT_CHILD_NONE) continue;
        bool moved;
        if (code >= 0) moved = dirty[code] != 0;
        else {
            uint32_t v = (uint32_t)(~code), first = v >> 3, cnt = (v & 7u) + 1u;
            moved = false;
            for (uint32_t k = first; k < first + cnt; ++k) { uint32_t m = recs[k].mesh; moved = moved || (m >= mesh_lo && m < mesh_hi); }
        }
        if (!moved) continue;
        box_of_child(code, nodes, recs, tris, verts, lo, hi);
        float *dl = side ? N.lo1 : N.lo0, *dh = side ? N.hi1 : N.hi0;
        for (int k = 0; k < 3; ++k) { dl[k] = lo[k]; dh[k] = hi[k]; }
        any = true;
    }
    dirty[ni] = any ? 1 : 0;
}
// The same for a SHORT list of nodes (the host walks its copy of the tree's topology down to the moved meshes): ONE
// launch of one workgroup that goes through the levels itself, deepest first, with a barrier between them -- a chain of
// ~30 launches of a few threads each costs more than the work.  list = [node ids, grouped by level | wide node ids];
// level_end[l] = end of group l in the list.  Every listed node recomputes both of its boxes (an unmoved side gets the
// values it already has).  Afterwards the listed wide nodes are re-quantised.
struct RtRefitLevels { uint32_t n_levels; uint32_t end[RT_REFIT_LIST_MAX_LEVELS]; };
__global__ __launch_bounds__(1024) void k_refit_list(RtBvhNode *nodes, RtTriRecord *recs, const rt_triangle *__restrict__ tris,
                                                      const rt_vertex *__restrict__ verts, const uint32_t *__restrict__ list,
                                                      RtRefitLevels lv, const RtWideSrc *__restrict__ src, RtWideNode *wide,
                                                      uint32_t n_wide_listed) {
    uint32_t begin = 0;
    for (uint32_t l = 0; l < lv.n_levels; ++l) {
        const uint32_t end = lv.end[l];
        for (uint32_t i = begin + threadIdx.x; i < end; i += blockDim.x) {
            RtBvhNode &N = nodes[list[i]];
            float lo[3], hi[3];
            if (N.child0 != RT_CHILD_NONE) {
                box_of_child(N.child0, nodes, recs, tris, verts, lo, hi);
                for (int k = 0; k < 3; ++k) { N.lo0[k] = lo[k]; N.hi0[k] = hi[k]; }
            }
            if (N.child1 != RT_CHILD_NONE) {
                box_of_child(N.child1, nodes, recs, tris, verts, lo, hi);
                for (int k = 0; k < 3; ++k) { N.lo1[k] = lo[k]; N.hi1[k] = hi[k]; }
            }
        }
        begin = end;
        __threadfence_block();    // the next level reads these boxes from other lanes' stores (one workgroup: its scope suffices)
        __syncthreads();
    }
    for (uint32_t i = threadIdx.x; i < n_wide_listed; i += blockDim.x) {
        const uint32_t w = list[begin + i];
        RtWideNode W = wide[w];
        rt_wide_quantise(nodes, src[w], W);
        wide[w] = W;
    }
}
hipError_t rtk_launch_refit_list(hipStream_t st, RtBvhNode *nodes, RtTriRecord *recs, const rt_triangle *tris, const rt_vertex *verts,
                                 const uint32_t *list, const uint32_t *level_end, uint32_t n_levels,
                                 const RtWideSrc *src, RtWideNode *wide, uint32_t n_wide_listed) {
    if (n_levels == 0 || n_levels > RT_REFIT_LIST_MAX_LEVELS) return n_levels ? hipErrorInvalidValue : hipSuccess;
    RtRefitLevels lv;
    lv.n_levels = n_levels;
    for (uint32_t l = 0; l < RT_REFIT_LIST_MAX_LEVELS; ++l) lv.end[l] = l < n_levels ? level_end[l] : 0u;
    hipLaunchKernelGGL(k_refit_list, dim3(1), dim3(1024), 0, st, nodes, recs, tris, verts, list, lv, src, wide, n_wide_listed);
    return hipGetLastError();
}
hipError_t rtk_launch_refit_level_dirty(hipStream_t st, RtBvhNode *nodes, RtTriRecord *recs, const rt_triangle *tris,
                                        const rt_vertex *verts, const uint32_t *level_nodes, uint32_t n,
                                        uint32_t mesh_lo, uint32_t mesh_hi, uint8_t *dirty) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_refit_level_dirty, dim3((n + 255) / 256), dim3(256), 0, st, nodes, recs, tris, verts, level_nodes, n, mesh_lo, mesh_hi, dirty);
    return hipGetLastError();
}
hipError_t rtk_launch_refit_level(hipStream_t st, RtBvhNode *nodes, RtTriRecord *recs, const rt_triangle *tris,
                                  const rt_vertex *verts, const uint32_t *level_nodes, uint32_t n) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_refit_level, dim3((n + 255) / 256), dim3(256), 0, st, nodes, recs, tris, verts, level_nodes, n);
    return hipGetLastError();
}

__global__ void k_wide_requant(const RtBvhNode *__restrict__ nodes, const RtWideSrc *__restrict__ src, RtWideNode *wide, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    RtWideNode W = wide[i];
    rt_wide_quantise(nodes, src[i], W);
    wide[i] = W;
}
__global__ void k_wide_requant_dirty(const RtBvhNode *__restrict__ nodes, const RtWideSrc *__restrict__ src, RtWideNode *wide, uint32_t n,
                                     const uint8_t *__restrict__ dirty) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const RtWideSrc S = src[i];
    bool any = false;       // a slot's box lives in binary node slot >> 1, which is dirty iff one of its two boxes changed
    for (int k = 0; k < 4; ++k) any = any || (S.slot[k] != RT_WIDE_EMPTY && dirty[S.slot[k] >> 1] != 0);
    if (!any) return;
    RtWideNode W = wide[i];
    rt_wide_quantise(nodes, S, W);
    wide[i] = W;
}
hipError_t rtk_launch_wide_requant_dirty(hipStream_t st, const RtBvhNode *nodes, const RtWideSrc *src, RtWideNode *wide, uint32_t n,
                                         const uint8_t *dirty) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_wide_requant_dirty, dim3((n + 255) / 256), dim3(256), 0, st, nodes, src, wide, n, dirty);
    return hipGetLastError();
}
__global__ void k_wide_requant_list(const RtBvhNode *__restrict__ nodes, const RtWideSrc *__restrict__ src, RtWideNode *wide,
                                    const uint32_t *__restrict__ list, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t w = list[i];
    RtWideNode W = wide[w];
    rt_wide_quantise(nodes, src[w], W);
    wide[w] = W;
}
hipError_t rtk_launch_wide_requant_list(hipStream_t st, const RtBvhNode *nodes, const RtWideSrc *src, RtWideNode *wide, const uint32_t *list, uint32_t n) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_wide_requant_list, dim3((n + 255) / 256), dim3(256), 0, st, nodes, src, wide, list, n);
    return hipGetLastError();
}
hipError_t rtk_launch_wide_requant(hipStream_t st, const RtBvhNode *nodes, const RtWideSrc *src, RtWideNode *wide, uint32_t n) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_wide_requant, dim3((n + 255) / 256), dim3(256), 0, st, nodes, src, wide, n);
    return hipGetLastError();
}

// running sums += the batch's per-sample contributions, one sample after the other (the order the reference's
// loop -- and the oracle -- adds them in); alpha counts samples
__global__ void k_accumulate_batch(float4 *__restrict__ accum, const float4 *__restrict__ contrib, size_t n_px, uint32_t batch) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n_px; i += stride) {
        float4 a = accum[i];
        for (uint32_t s = 0; s < batch; ++s) {
            float4 l = contrib[(size_t)s * n_px + i];     // w: 1 where the launch covered the pixel (viewport, shard), else 0
            a.w = a.w + l.w;
            a.x = a.x + l.x; a.y = a.y + l.y; a.z = a.z + l.z;
        }
        accum[i] = a;
    }
}

// ---- A8 resolve: light / f32(samples) (ray_tracing.wgsl:212) ----------------------------------------
__global__ void k_resolve(const float4 *__restrict__ accum, float4 *__restrict__ rgba, size_t n, float ns) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        float4 a = accum[i];
        rgba[i] = make_float4(a.x / ns, a.y / ns, a.z / ns, a.w / ns);
    }
}

// ---- presentation: clamp + sRGB encode to BGRA8 (src/pipeline.rs:56-60) ------------------------------
// Byte work, so it is exact: a colour byte = round(255 * srgb_oetf(x)) = the number of linear-space thresholds <= x
// (rt_srgb_table.h: for k = 1..255 the smallest binary32 x whose byte is >= k, exactly rounded from the OETF by
// tools/gen_srgb_table.py).  Float compares only -- the oracle counts over the same values and the bytes are identical
// (a powf here against a pow in double there differed by 1 LSB on ~1 % of the bytes).  NaN and x <= 0 give 0, x >= 1
// gives 255.  The table sits in LDS (1 KiB per block): eight dependent ds_reads per channel, no divergent constant fetch.
#define RT_SRGB_TABLE_QUAL static __device__ const
#include "rt_srgb_table.h"
__device__ __forceinline__ uint32_t srgb_byte(const float *tab, float x) {
    uint32_t n = 0;
#pragma unroll
    for (uint32_t step = 128u; step; step >>= 1) n += tab[n + step - 1u] <= x ? step : 0u;
    return n;
}
__device__ __forceinline__ uint32_t unorm8(float x) {
    if (!(x > 0.0f)) x = 0.0f;
    if (x > 1.0f) x = 1.0f;
    return (uint32_t)floorf(x * 255.0f + 0.5f);
}
__global__ void k_encode_bgra8(const float4 *__restrict__ rgba, uint32_t *__restrict__ bgra, size_t n) {
    __shared__ float tab[256];
    for (uint32_t k = threadIdx.x; k < 256u; k += blockDim.x) tab[k] = k < 255u ? __uint_as_float(RT_SRGB_THRESHOLD_BITS[k]) : RT_INF;
    __syncthreads();
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        float4 c = rgba[i];
        bgra[i] = srgb_byte(tab, c.z) | (srgb_byte(tab, c.y) << 8) | (srgb_byte(tab, c.x) << 16) | (unorm8(c.w) << 24);
    }
}


// ---- multi-GPU tile exchange: this rank's shard tiles, packed ---------------------------------------
// Rank r owns the shard tiles g = r + k * count (k = 0, 1, ...: its owned-tile ordinal; g = ty * tiles_x + tx over the
// covered rect, as pixel_of_thread / pixel_of_region deal them).  Packed layout: tile k occupies tile*tile float4
// slots, row-major inside the tile; slots outside the covered rect (edge tiles) are written as zeros and ignored by
// the unpack.  One thread per float4, consecutive threads = consecutive pixels of a tile row: both sides coalesce.
__global__ void k_tile_pack(const float4 *__restrict__ frame, float4 *__restrict__ packed, RtViewDev V, RtShardDev sh) {
    const uint32_t per_tile = sh.tile * sh.tile;
    const size_t n = (size_t)sh.n_owned * per_tile;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const uint32_t k = (uint32_t)(i / per_tile), w = (uint32_t)(i % per_tile);
        const uint32_t g = sh.rank + k * sh.count;
        const uint32_t x = (g % sh.tiles_x) * sh.tile + w % sh.tile, y = (g / sh.tiles_x) * sh.tile + w / sh.tile;
        float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (x < V.vw && y < V.vh) v = frame[(size_t)(V.vy + y) * V.width + (V.vx + x)];
        packed[i] = v;
    }
}
// The other direction, for n_src ranks at once: slot s of `packed` (slot_stride float4 apart) holds the packed tiles
// of rank first_rank + s; every pixel of those tiles goes to its place in `frame`.
__global__ void k_tile_unpack(float4 *__restrict__ frame, const float4 *__restrict__ packed, size_t slot_stride, RtViewDev V,
                              uint32_t tile, uint32_t tiles_x, uint32_t n_tiles, uint32_t count, uint32_t first_rank, uint32_t n_src) {
    const uint32_t per_tile = tile * tile;
    const uint32_t max_owned = (n_tiles + count - 1u) / count;
    const size_t per_rank = (size_t)max_owned * per_tile;
    const size_t n = per_rank * n_src;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const uint32_t s = (uint32_t)(i / per_rank);
        const size_t j = i % per_rank;
        const uint32_t k = (uint32_t)(j / per_tile), w = (uint32_t)(j % per_tile);
        const uint32_t g = first_rank + s + k * count;
        if (g >= n_tiles) continue;
        const uint32_t x = (g % tiles_x) * tile + w % tile, y = (g / tiles_x) * tile + w / tile;
        if (x < V.vw && y < V.vh) frame[(size_t)(V.vy + y) * V.width + (V.vx + x)] = packed[(size_t)s * slot_stride + j];
    }
}

__global__ void k_fill_u32(uint32_t *p, uint32_t v, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = v;
}

// ---- launchers --------------------------------------------------------------------------------------
uint32_t rtk_primary_blocks(const RtShardDev &sh) {
    uint32_t sub = sh.tile / RT_WORK_TILE;
    return ((sh.n_owned + 7u) / 8u) * 8u * sub * sub;   // tiles rounded up to a multiple of 8 (XCD rotation)
}

hipError_t rtk_launch_primary(hipStream_t st, const RtSceneDev &S, const RtViewDev &V, const RtShardDev &sh,
                              uint32_t flags, float *rgba, uint32_t *prim, float *t, unsigned long long *counters,
                              const RtTune &tune) {
    uint32_t nb = rtk_primary_blocks(sh);
    if (nb == 0) return hipSuccess;
    bool count = (flags & RT_FLAG_COUNTERS) != 0;
    bool packet = (flags & RT_FLAG_WAVE_TRAVERSAL) && !(flags & RT_FLAG_BRUTE_FORCE);
    if (!packet && !(flags & RT_FLAG_STATIC_GRID) && !(flags & RT_FLAG_BRUTE_FORCE)) {
        // persistent kernel: a grid that fills the chip once; blocks that start late find the queues
        // empty and exit, so no residency assumption is made
        // Waves in flight: a tree that misses the caches needs all 8 waves per SIMD to hide the misses (85 MB of nodes
        // and records: 0.74 ms with 8 blocks per CU, 0.95 with 3); a tree that one XCD's 4 MiB L2 holds does not, and
        // fewer waves each run faster and leave a shorter tail at the end of the frame (the reference's default scene
        // at 1080p: 0.27 ms with 3 blocks per CU, 0.37 with 8; tools/occupancy_sweep.py).
        uint32_t per_cu = tune.blocks_per_cu;
        const bool w_ = S.n_wide != 0u && !(flags & RT_FLAG_BINARY_TREE);
        const size_t tree = (size_t)(w_ ? S.n_wide : S.n_nodes) * sizeof(RtBvhNode) + (size_t)S.n_recs * sizeof(RtTriRecord);
        // With 3-4 frames in flight a CU holds blocks of several launches; 5 per launch instead of 8 leave room for the next
        // frames' blocks and each launch a shorter tail (tools/ab_in_flight_knobs.py, 4 in flight: bench scene 0.603 -> 0.597 ms,
        // 1 M-triangle soup 0.412 -> 0.396; one frame at a time the same setting costs 10 %)
        // A SHARD of a frame (tiles mode at N >= 4: at most 8 blocks per CU of work in the whole launch) with frames in flight: 3 per
        // launch, so that the launches in flight are resident together and a wave takes more than one region
        // (profiles/r3/shard_time_blocks.txt, one GPU playing one rank of N = 4 / 8, 4 in flight: 0.156 -> 0.151 / 0.096 -> 0.092 ms)
        const uint32_t need_blocks = (sh.n_owned * sh.tile * sh.tile + RT_BLOCK - 1) / RT_BLOCK;
        const uint32_t in_flight_per_cu = need_blocks <= 8u * tune.cus ? 3u : 5u;
        if (per_cu == 0) per_cu = tree <= ((size_t)4 << 20) ? 3u : (tune.frames_in_flight >= 3u ? in_flight_per_cu : (tree <= ((size_t)16 << 20) ? 5u : 8u));
        // 8 waves per SIMD with frames in flight (k_stream8), else the 7-wave build; tune.stream_waves (7 / 8) overrides for A/Bs
        const bool eight = tune.stream_waves ? tune.stream_waves == 8u : tune.frames_in_flight >= 3u;
        const uint32_t cu_holds = eight ? 8u : (uint32_t)RT_STREAM_WAVES;
        if (per_cu > cu_holds) per_cu = cu_holds;      // what a CU can hold: later blocks would only find empty queues
        const uint32_t need = need_blocks;
        uint32_t grid = tune.cus * per_cu < need ? tune.cus * per_cu : need;
        if (grid == 0) return hipSuccess;
        const bool wide = S.n_wide != 0u && !(flags & RT_FLAG_BINARY_TREE);
        RtStreamTune tn;
        tn.refill_min = tune.refill_min; tn.finish_min = tune.finish_min; tn.uni_min = tune.uni_min;
        tn.vote_node = wide ? tune.wide_vote_node : tune.vote_node; tn.vote_tri = wide ? tune.wide_vote_tri : tune.vote_tri;
#define RT_LAUNCH_STREAM(K, C, W) \
    hipLaunchKernelGGL((K<C, W>), dim3(grid), dim3(RT_BLOCK), 0, st, S, V, sh, flags, tn, (float4 *)rgba, prim, t, counters)
        if (eight) {
            if (count) { if (wide) RT_LAUNCH_STREAM(k_stream8, true, true); else RT_LAUNCH_STREAM(k_stream8, true, false); }
            else { if (wide) RT_LAUNCH_STREAM(k_stream8, false, true); else RT_LAUNCH_STREAM(k_stream8, false, false); }
        } else {
            if (count) { if (wide) RT_LAUNCH_STREAM(k_stream, true, true); else RT_LAUNCH_STREAM(k_stream, true, false); }
            else { if (wide) RT_LAUNCH_STREAM(k_stream, false, true); else RT_LAUNCH_STREAM(k_stream, false, false); }
        }
#undef RT_LAUNCH_STREAM
        return hipGetLastError();
    }
#define RT_LAUNCH_PRIMARY(C, P) \
    hipLaunchKernelGGL((k_primary<C, P>), dim3(nb), dim3(RT_BLOCK), 0, st, S, V, sh, nb, flags, (float4 *)rgba, prim, t, counters)
    if (count && packet) RT_LAUNCH_PRIMARY(true, true);
    else if (count) RT_LAUNCH_PRIMARY(true, false);
    else if (packet) RT_LAUNCH_PRIMARY(false, true);
    else RT_LAUNCH_PRIMARY(false, false);
#undef RT_LAUNCH_PRIMARY
    return hipGetLastError();
}

hipError_t rtk_launch_path_start(hipStream_t st, const RtSceneDev &S, const RtViewDev &V, const RtShardDev &sh,
                                 uint32_t flags, uint32_t sample, uint32_t seed, uint32_t stream_id,
                                 uint32_t max_segments, float *contrib, uint32_t n_px, uint32_t kshift, RtPathQueue qout,
                                 unsigned long long *counters, int qout_counter, const RtTune &tune) {
    RtShardDev shs = sh;
    shs.tile <<= kshift;
    uint32_t nb = rtk_primary_blocks(shs);      // 16x16 blocks of the magnified grid
    if (nb == 0) return hipSuccess;
    uint32_t persist = (flags & (RT_FLAG_STATIC_GRID | RT_FLAG_BRUTE_FORCE)) ? 0u : 1u;
    uint32_t grid = persist && tune.persist_blocks < nb ? (tune.persist_blocks & ~7u) : nb;
    if (grid == 0) grid = nb;
    if (flags & RT_FLAG_COUNTERS)
        hipLaunchKernelGGL(k_path_start<true>, dim3(grid), dim3(RT_BLOCK), 0, st, S, V, sh, nb, flags, sample, seed, stream_id, max_segments, (float4 *)contrib, n_px, kshift, qout, counters, qout_counter, persist);
    else
        hipLaunchKernelGGL(k_path_start<false>, dim3(grid), dim3(RT_BLOCK), 0, st, S, V, sh, nb, flags, sample, seed, stream_id, max_segments, (float4 *)contrib, n_px, kshift, qout, counters, qout_counter, persist);
    return hipGetLastError();
}

hipError_t rtk_launch_path_bounce(hipStream_t st, const RtSceneDev &S, uint32_t flags, uint32_t capacity,
                                  int is_last, float *contrib, uint32_t n_px, RtPathQueue qin, RtPathQueue qout,
                                  unsigned long long *counters, int qin_counter, int qout_counter, const RtTune &tune) {
    uint32_t nb = (capacity + RT_BLOCK - 1) / RT_BLOCK;
    if (nb == 0) return hipSuccess;
    uint32_t persist = (flags & (RT_FLAG_STATIC_GRID | RT_FLAG_BRUTE_FORCE)) ? 0u : 1u;
    uint32_t grid = persist && tune.persist_blocks < nb ? tune.persist_blocks : nb;
    if (flags & RT_FLAG_COUNTERS)
        hipLaunchKernelGGL(k_path_bounce<true>, dim3(grid), dim3(RT_BLOCK), 0, st, S, flags, is_last, (float4 *)contrib, n_px, qin, qout, counters, qin_counter, qout_counter, persist);
    else
        hipLaunchKernelGGL(k_path_bounce<false>, dim3(grid), dim3(RT_BLOCK), 0, st, S, flags, is_last, (float4 *)contrib, n_px, qin, qout, counters, qin_counter, qout_counter, persist);
    return hipGetLastError();
}


static uint32_t stream_grid(size_t n) {
    size_t b = (n + 255) / 256;
    return (uint32_t)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}
hipError_t rtk_launch_resolve(hipStream_t st, const float *accum, float *rgba, size_t n_px, float, uint32_t total_samples) {
    if (n_px == 0) return hipSuccess;
    hipLaunchKernelGGL(k_resolve, dim3(stream_grid(n_px)), dim3(256), 0, st, (const float4 *)accum, (float4 *)rgba, n_px, (float)total_samples);
    return hipGetLastError();
}
hipError_t rtk_launch_accumulate_batch(hipStream_t st, float *accum, const float *contrib, size_t n_px, uint32_t batch) {
    if (n_px == 0 || batch == 0) return hipSuccess;
    hipLaunchKernelGGL(k_accumulate_batch, dim3(stream_grid(n_px)), dim3(256), 0, st, (float4 *)accum, (const float4 *)contrib, n_px, batch);
    return hipGetLastError();
}
hipError_t rtk_launch_encode_bgra8(hipStream_t st, const float *rgba, uint8_t *bgra, size_t n_px) {
    if (n_px == 0) return hipSuccess;
    hipLaunchKernelGGL(k_encode_bgra8, dim3(stream_grid(n_px)), dim3(256), 0, st, (const float4 *)rgba, (uint32_t *)bgra, n_px);
    return hipGetLastError();
}

hipError_t rtk_launch_tile_pack(hipStream_t st, const float *frame, float *packed, const RtViewDev &V, const RtShardDev &sh) {
    const size_t n = (size_t)sh.n_owned * sh.tile * sh.tile;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_tile_pack, dim3(stream_grid(n)), dim3(256), 0, st, (const float4 *)frame, (float4 *)packed, V, sh);
    return hipGetLastError();
}
hipError_t rtk_launch_tile_unpack(hipStream_t st, float *frame, const float *packed, size_t slot_stride_px, const RtViewDev &V,
                                  uint32_t tile, uint32_t tiles_x, uint32_t n_tiles, uint32_t count, uint32_t first_rank, uint32_t n_src) {
    const size_t n = (size_t)((n_tiles + count - 1u) / count) * tile * tile * n_src;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_tile_unpack, dim3(stream_grid(n)), dim3(256), 0, st, (float4 *)frame, (const float4 *)packed, slot_stride_px, V,
                       tile, tiles_x, n_tiles, count, first_rank, n_src);
    return hipGetLastError();
}

// ---- static scratch for rt_render_static (callers that cannot allocate: include/rt_amd.h) -------------------------
// The work-queue heads, counters, path queues and contribution plane of ONE frame, in a file-scope device array instead
// of hipMalloc'ed memory.  Every GPU the library runs on holds its own copy at its own address for the life of the
// process; `rt_static_scratch` below is only the host-side handle of the symbol -- host code never passes it to a kernel,
// it asks each device for the array's address once (rtk_static_scratch_address, called by rt_create with that device
// current) and uses that pointer from then on.
static __device__ __align__(256) unsigned char rt_static_scratch[RT_STATIC_SCRATCH_BYTES];
hipError_t rtk_static_scratch_address(void **ptr) {
    void *p = nullptr;
    hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(rt_static_scratch));
    if (e != hipSuccess) return e;
    if (!p) return hipErrorInvalidSymbol;
    *ptr = p;
    return hipSuccess;
}
hipError_t rtk_launch_fill_u32(hipStream_t st, uint32_t *p, uint32_t value, size_t n) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_fill_u32, dim3(stream_grid(n)), dim3(256), 0, st, p, value, n);
    return hipGetLastError();
}
