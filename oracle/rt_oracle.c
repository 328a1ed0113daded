/*
 * rt_oracle.c -- CPU ORACLE (test infrastructure; see rt_oracle.h for scope and pinning).
 * Plain C99 + pthreads.  Build: oracle/Makefile (gcc -O2 -ffp-contract=off).
 * Every function names the reference file:line it restates.  "parity unpinned" by the
 * reference; pinned by KAT-1..6 and self-generated golden images.
 */
#include "rt_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ---- float helpers: fixed operation order (see header) ------------------------------ */
typedef struct { float x, y, z; } v3;

static inline v3 V(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 ld3(const float *p) { return V(p[0], p[1], p[2]); }
static inline void st3(float *p, v3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }
static inline v3 sub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 add(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 mul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 scl(v3 a, float s) { return V(a.x * s, a.y * s, a.z * s); }
static inline float dot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline v3 cross(v3 a, v3 b) {
    return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline v3 normalize(v3 a) {
    float len = sqrtf(dot(a, a));
    return V(a.x / len, a.y / len, a.z / len);
}

/* ---- A1 ray generation: wgsl:174,176-177 ---------------------------------------------- */
void orc_ray_gen(const orc_view *v, uint32_t px, uint32_t py, float o[3], float d[3]) {
    /* in.uv of Bevy's fullscreen triangle = pixel centre / target size, relative to the
     * viewport when one is set (src/node.rs:71-73). */
    uint32_t vw = v->vp_w ? v->vp_w : v->width, vh = v->vp_w ? v->vp_h : v->height;
    uint32_t vx = v->vp_w ? v->vp_x : 0, vy = v->vp_w ? v->vp_y : 0;
    float u = ((float)(px - vx) + 0.5f) / (float)vw;
    float w = ((float)(py - vy) + 0.5f) / (float)vh;
    /* uv = in.uv * vec2(2,-2) + vec2(-1,1)   (wgsl:174) */
    float nx = u * 2.0f + -1.0f;
    float ny = w * -2.0f + 1.0f;
    const float *m = v->world_from_clip; /* column-major */
    /* world_from_clip * vec4(uv, 0, 1) = ((c0*x + c1*y) + c2*0) + c3*1   (wgsl:177) */
    float r[4];
    for (int i = 0; i < 4; ++i)
        r[i] = ((m[0 + i] * nx + m[4 + i] * ny) + m[8 + i] * 0.0f) + m[12 + i] * 1.0f;
    /* normalize over the vec4, then .xyz */
    float len = sqrtf(((r[0] * r[0] + r[1] * r[1]) + r[2] * r[2]) + r[3] * r[3]);
    d[0] = r[0] / len; d[1] = r[1] / len; d[2] = r[2] / len;
    o[0] = v->world_position[0]; o[1] = v->world_position[1]; o[2] = v->world_position[2];
}

/* ---- A3 ray/AABB cull: wgsl:141-148 ----------------------------------------------------- */
int orc_ray_aabb(const float o[3], const float d[3], const float lb[3], const float rt[3]) {
    v3 inv = V(1.0f / d[0], 1.0f / d[1], 1.0f / d[2]);
    v3 t1 = mul(sub(ld3(lb), ld3(o)), inv);
    v3 t2 = mul(sub(ld3(rt), ld3(o)), inv);
    float tmin = fmaxf(fmaxf(fminf(t1.x, t2.x), fminf(t1.y, t2.y)), fminf(t1.z, t2.z));
    float tmax = fminf(fminf(fmaxf(t1.x, t2.x), fmaxf(t1.y, t2.y)), fmaxf(t1.z, t2.z));
    return tmax >= tmin; /* no tmax>=0 test, no clamp (KAT-4) */
}

/* ---- A4 ray/triangle: wgsl:118-139 ------------------------------------------------------ */
void orc_ray_triangle(const orc_scene *s, const float o[3], const float d[3], uint32_t tri, orc_hit *out) {
    const orc_triangle *T = &s->tris[tri];
    const orc_vertex *va = &s->verts[T->idx[0]], *vb = &s->verts[T->idx[1]], *vc = &s->verts[T->idx[2]];
    v3 ro = ld3(o), rd = ld3(d);
    v3 edge_ab = sub(ld3(vb->pos), ld3(va->pos));
    v3 edge_ac = sub(ld3(vc->pos), ld3(va->pos));
    v3 norm = cross(edge_ab, edge_ac);
    /* ray_cross (wgsl:125) is computed and unused in the reference; omitted */
    v3 ao = sub(ro, ld3(va->pos));
    v3 dao = cross(ao, rd);

    float det = -dot(rd, norm);
    float inv_det = 1.0f / det;

    float dst = dot(ao, norm) * inv_det;
    float u = dot(edge_ac, dao) * inv_det;
    float v = -dot(edge_ab, dao) * inv_det;
    float w = 1.0f - u - v;

    out->hit = det >= 1e-6f && dst >= 0.0f && u >= 0.0f && v >= 0.0f && w >= 0.0f;
    st3(out->point, add(ro, scl(rd, dst)));
    /* normalize(a.norm*w + b.norm*u + c.norm*v): u weights B, v weights C (KAT-5) */
    v3 n = add(add(scl(ld3(va->norm), w), scl(ld3(vb->norm), u)), scl(ld3(vc->norm), v));
    st3(out->normal, normalize(n));
    out->t = dst;
    out->material = 0;
    out->prim = tri;
}

/* ---- extension: ray / analytic sphere (build-defined; see rt_oracle.h) ----------------------- */
void orc_ray_sphere(const orc_scene *s, const float o[3], const float d[3], uint32_t sphere, orc_hit *out) {
    const orc_sphere *S = &s->spheres[sphere];
    v3 ro = ld3(o), rd = ld3(d), c = ld3(S->center);
    v3 oc = sub(ro, c);
    float a = dot(rd, rd);
    float hb = dot(oc, rd);
    float cc = dot(oc, oc) - S->radius * S->radius;
    float disc = hb * hb - a * cc;
    float dst = (-hb - sqrtf(disc)) / a;
    out->hit = disc >= 0.0f && dst >= 0.0f;
    v3 p = add(ro, scl(rd, dst));
    st3(out->point, p);
    st3(out->normal, normalize(sub(p, c)));
    out->t = dst;
    out->material = s->sphere_mats ? s->sphere_mats[sphere] : 0u;
    out->prim = (uint32_t)s->n_tri + sphere;
}

/* ---- A5 closest hit over meshes then triangles: wgsl:150-170 --------------------------- */
void orc_hit_triangles(const orc_scene *s, const float o[3], const float d[3], uint32_t flags, orc_hit *out) {
    orc_hit hit;
    memset(&hit, 0, sizeof hit); /* no_hit(): wgsl:89-91 */
    hit.prim = ORC_MISS_PRIM;
    for (size_t i = 0; i < s->n_mesh; ++i) {
        const orc_mesh_info *mesh = &s->meshes[i];
        if (!(flags & ORC_FLAG_NO_MESH_CULL) && !orc_ray_aabb(o, d, mesh->aabb_min, mesh->aabb_max))
            continue;
        for (uint32_t j = mesh->first_tri; j < mesh->first_tri + mesh->tri_count; ++j) {
            orc_hit rec;
            orc_ray_triangle(s, o, d, j, &rec);
            rec.material = mesh->material;
            if (!rec.hit) continue;
            if (!hit.hit || hit.t > rec.t) hit = rec; /* strict: earlier wins ties (KAT-3) */
        }
    }
    for (size_t i = 0; i < s->n_sphere; ++i) { /* extension: spheres come after every mesh in loop order */
        orc_hit rec;
        orc_ray_sphere(s, o, d, (uint32_t)i, &rec);
        if (!rec.hit) continue;
        if (!hit.hit || hit.t > rec.t) hit = rec;
    }
    *out = hit;
}

/* ---- tree walk: scalar restatement of the product's traversal (no reference counterpart; rt_oracle.h) ------ */
#define TW_NONE ((int32_t)0x80000000)
#define TW_DONE ((int32_t)0x80000001)
#define TW_STACK 256
#define TW_INV_MAX 1e30f
typedef struct { float lo0[3], hi0[3], lo1[3], hi1[3]; int32_t child0, child1; uint32_t spare[2]; } tw_node;
typedef struct { float org[3], scl[3]; uint32_t lo[3], hi[3]; int32_t child[4]; } tw_wide;
typedef struct { float a[3], ab[3], ac[3]; uint32_t tri, mesh, seq; } tw_rec;
typedef struct { int32_t e[TW_STACK]; int sp; } tw_stack;

static inline int32_t tw_pop(tw_stack *st) { return st->sp == 0 ? TW_DONE : st->e[--st->sp]; }
static inline void tw_push(tw_stack *st, int32_t v) { if (st->sp < TW_STACK) st->e[st->sp] = v; ++st->sp; }
static inline float tw_ub(uint32_t w, int k) { return (float)((w >> (8 * k)) & 0xFFu); }

/* binary step: both child boxes, nearer first, the other pushed; far distances get 2 ulp of slack */
static inline int32_t tw_step_binary(const tw_node *N, v3 o, v3 inv, float t, tw_stack *st) {
    const float limit = t * 1.00001f;
    float ax0 = (N->lo0[0] - o.x) * inv.x, ax1 = (N->hi0[0] - o.x) * inv.x;
    float ay0 = (N->lo0[1] - o.y) * inv.y, ay1 = (N->hi0[1] - o.y) * inv.y;
    float az0 = (N->lo0[2] - o.z) * inv.z, az1 = (N->hi0[2] - o.z) * inv.z;
    float bx0 = (N->lo1[0] - o.x) * inv.x, bx1 = (N->hi1[0] - o.x) * inv.x;
    float by0 = (N->lo1[1] - o.y) * inv.y, by1 = (N->hi1[1] - o.y) * inv.y;
    float bz0 = (N->lo1[2] - o.z) * inv.z, bz1 = (N->hi1[2] - o.z) * inv.z;
    float tmin0 = fmaxf(fmaxf(fminf(ax0, ax1), fminf(ay0, ay1)), fminf(az0, az1));
    float tmax0 = fminf(fminf(fmaxf(ax0, ax1), fmaxf(ay0, ay1)), fmaxf(az0, az1));
    float tmin1 = fmaxf(fmaxf(fminf(bx0, bx1), fminf(by0, by1)), fminf(bz0, bz1));
    float tmax1 = fminf(fminf(fmaxf(bx0, bx1), fmaxf(by0, by1)), fmaxf(bz0, bz1));
    tmax0 *= 1.0000005f; tmax1 *= 1.0000005f;
    int h0 = tmin0 <= tmax0 && tmax0 >= 0.0f && tmin0 <= limit;
    int h1 = tmin1 <= tmax1 && tmax1 >= 0.0f && tmin1 <= limit;
    int both = h0 && h1, swap = tmin1 < tmin0;
    if (both) tw_push(st, swap ? N->child0 : N->child1);
    if (h0 || h1) return (both ? swap : h1) ? N->child1 : N->child0;
    return tw_pop(st);
}

/* wide step: plane = org + byte*scl, distance = fma(byte, A, -B) with A = scl*inv, B = (o - org)*inv and 1/d clamped
 * to +-1e30; near/far bytes by the sign of 1/d; entry distance clamped to +0; far distance *1.000001, then the limit;
 * nearest child exactly (three compare-exchanges on the distance bits), the others pushed in the order they end in */
static inline __attribute__((always_inline)) int32_t tw_step_wide(const tw_wide *W, v3 o, v3 inv, float t, tw_stack *st) {
    const float limit = t * 1.00001f;
    const float Ax = W->scl[0] * inv.x, Ay = W->scl[1] * inv.y, Az = W->scl[2] * inv.z;
    const float Bx = (o.x - W->org[0]) * inv.x, By = (o.y - W->org[1]) * inv.y, Bz = (o.z - W->org[2]) * inv.z;
    const int sx = inv.x < 0.0f, sy = inv.y < 0.0f, sz = inv.z < 0.0f;
    const uint32_t nx = sx ? W->hi[0] : W->lo[0], fx = sx ? W->lo[0] : W->hi[0];
    const uint32_t ny = sy ? W->hi[1] : W->lo[1], fy = sy ? W->lo[1] : W->hi[1];
    const uint32_t nz = sz ? W->hi[2] : W->lo[2], fz = sz ? W->lo[2] : W->hi[2];
    uint32_t key[4]; int32_t ref[4];
    for (int k = 0; k < 4; ++k) {
        float tn = fmaxf(fmaxf(__builtin_fmaf(tw_ub(nx, k), Ax, -Bx), __builtin_fmaf(tw_ub(ny, k), Ay, -By)), __builtin_fmaf(tw_ub(nz, k), Az, -Bz));
        float tf = fminf(fminf(__builtin_fmaf(tw_ub(fx, k), Ax, -Bx), __builtin_fmaf(tw_ub(fy, k), Ay, -By)), __builtin_fmaf(tw_ub(fz, k), Az, -Bz));
        tn = tn > 0.0f ? tn : 0.0f;           /* max(tn, +0): -0 and NaN become +0, so the bits order like the value */
        tf = fminf(tf * 1.000001f, limit);
        int hit = tn <= tf;
        uint32_t bits; memcpy(&bits, &tn, 4);
        key[k] = hit ? bits : 0xFFFFFFFFu;
        ref[k] = hit ? W->child[k] : TW_NONE;
    }
#define TW_CE(a, b) if (key[b] < key[a]) { uint32_t tk = key[a]; int32_t tr_ = ref[a]; key[a] = key[b]; ref[a] = ref[b]; key[b] = tk; ref[b] = tr_; }
    TW_CE(0, 1) TW_CE(2, 3) TW_CE(0, 2)
#undef TW_CE
    { uint32_t tk = key[1]; int32_t tr_ = ref[1]; key[1] = key[2]; ref[1] = ref[2]; key[2] = tk; ref[2] = tr_; }
    if (ref[3] != TW_NONE) tw_push(st, ref[3]);
    if (ref[2] != TW_NONE) tw_push(st, ref[2]);
    if (ref[1] != TW_NONE) tw_push(st, ref[1]);
    return ref[0] != TW_NONE ? ref[0] : tw_pop(st);
}

/* one leaf record against the ray: wgsl:118-139 from (a, ab, ac), or the build-defined sphere test */
static inline int tw_prim(const tw_rec *R, v3 o, v3 d, float *dst, float *u, float *v) {
    if (R->mesh == 0xFFFFFFFFu) {
        v3 oc = sub(o, ld3(R->a));
        float a = dot(d, d), hb = dot(oc, d), cc = dot(oc, oc) - R->ab[0] * R->ab[0];
        float disc = hb * hb - a * cc;
        *dst = (-hb - sqrtf(disc)) / a; *u = 0.0f; *v = 0.0f;
        return disc >= 0.0f && *dst >= 0.0f;
    }
    v3 ab = ld3(R->ab), ac = ld3(R->ac);
    v3 norm = cross(ab, ac), ao = sub(o, ld3(R->a)), dao = cross(ao, d);
    float det = -dot(d, norm), inv_det = 1.0f / det;
    *dst = dot(ao, norm) * inv_det;
    *u = dot(ac, dao) * inv_det;
    *v = -dot(ab, dao) * inv_det;
    float w = 1.0f - *u - *v;
    return det >= 1e-6f && *dst >= 0.0f && *u >= 0.0f && *v >= 0.0f && w >= 0.0f;
}

static inline __attribute__((always_inline)) void tw_walk(const orc_scene *s, const orc_tree *tr, const float o_[3], const float d_[3],
                                                          uint32_t flags, int any_hit, orc_hit *out, orc_walk_counts *cnt) {
    const int wide = tr->n_wide != 0;
    const tw_node *nodes = (const tw_node *)tr->nodes;
    const tw_wide *wnodes = (const tw_wide *)tr->wide;
    const tw_rec *recs = (const tw_rec *)tr->recs;
    v3 o = ld3(o_), d = ld3(d_);
    float t = INFINITY; uint32_t seq = 0xFFFFFFFFu, rec = 0xFFFFFFFFu;
    uint64_t n_nodes = 0, n_tris = 0, n_culls = 0;
    memset(out, 0, sizeof *out); out->prim = ORC_MISS_PRIM;
    /* a ray with a NaN in it misses everything in the reference's loop (every compare of wgsl:137 is false): the walk
     * ends it before it starts, as the kernels do (the wide step's min/max would drop the NaN and visit every node) */
    const int nan_ray = o.x != o.x || o.y != o.y || o.z != o.z || d.x != d.x || d.y != d.y || d.z != d.z;
    if ((wide ? tr->n_wide : tr->n_nodes) != 0 && !nan_ray) {
        v3 inv = V(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
        if (wide) inv = V(fminf(fmaxf(inv.x, -TW_INV_MAX), TW_INV_MAX), fminf(fmaxf(inv.y, -TW_INV_MAX), TW_INV_MAX),
                          fminf(fmaxf(inv.z, -TW_INV_MAX), TW_INV_MAX));
        tw_stack st; st.sp = 0;
        int32_t cur = 0;
        while (cur != TW_DONE) {
            if (cur >= 0) {
                ++n_nodes;
                cur = wide ? tw_step_wide(&wnodes[cur], o, inv, t, &st) : tw_step_binary(&nodes[cur], o, inv, t, &st);
            } else if (cur == TW_NONE) {
                cur = tw_pop(&st);
            } else {
                uint32_t code = (uint32_t)(~cur), first = code >> 3, left = code & 7u;
                const tw_rec *R = &recs[first];
                float dst, u, v;
                ++n_tris;
                int ok = tw_prim(R, o, d, &dst, &u, &v);
                ok = ok && (dst < t || (dst == t && R->seq < seq));
                if (ok && !(flags & ORC_FLAG_NO_MESH_CULL) && R->mesh != 0xFFFFFFFFu) {   /* per-mesh cull, wgsl:155: exact 1/d */
                    ok = orc_ray_aabb(o_, d_, s->meshes[R->mesh].aabb_min, s->meshes[R->mesh].aabb_max);
                    ++n_culls;
                }
                if (ok) { t = dst; seq = R->seq; rec = first; }
                if (ok && any_hit) cur = TW_DONE;
                else if (left == 0u) cur = tw_pop(&st);
                else cur = ~(int32_t)(((first + 1u) << 3) | (left - 1u));
            }
        }
    }
    if (rec != 0xFFFFFFFFu) {      /* point, normal, material of the accepted record: the reference's own arithmetic */
        const tw_rec *R = &recs[rec];
        if (R->mesh == 0xFFFFFFFFu) orc_ray_sphere(s, o_, d_, R->tri - (uint32_t)s->n_tri, out);
        else { orc_ray_triangle(s, o_, d_, R->tri, out); out->material = s->meshes[R->mesh].material; }
        out->hit = 1; out->t = t;
    }
    if (cnt) { cnt->node_visits += n_nodes; cnt->tri_tests += n_tris; cnt->mesh_culls += n_culls; }
}
/* the fma of the wide step as ONE instruction where the CPU has it (same result as libm's fmaf, much faster) */
#if defined(__x86_64__)
__attribute__((target("fma"))) static void tw_walk_fma(const orc_scene *s, const orc_tree *tr, const float o[3], const float d[3],
                                                       uint32_t flags, int any_hit, orc_hit *out, orc_walk_counts *cnt) {
    tw_walk(s, tr, o, d, flags, any_hit, out, cnt);
}
#endif
void orc_hit_tree(const orc_scene *s, const orc_tree *tr, const float o[3], const float d[3], uint32_t flags,
                  int any_hit, orc_hit *out, orc_walk_counts *cnt) {
#if defined(__x86_64__)
    if (__builtin_cpu_supports("fma")) { tw_walk_fma(s, tr, o, d, flags, any_hit, out, cnt); return; }
#endif
    tw_walk(s, tr, o, d, flags, any_hit, out, cnt);
}

/* ---- A6 shade: wgsl:203-211 --------------------------------------------------------------- */
void orc_shade(const orc_scene *s, const orc_hit *h, float rgba[4]) {
    if (h->hit) {
        if (h->material >= 100u) { /* debug grey, wgsl:205-207 */
            float g = 0.5f + (float)(h->material - 100u) / 6.0f;
            rgba[0] = rgba[1] = rgba[2] = rgba[3] = g;
            return;
        }
        float ndl = dot(ld3(h->normal), V(0.0f, 1.0f, 0.0f));
        /* out-of-range material (possible after the P1 desync, src/ray_tracing.rs:122-129):
         * WGSL robust buffer access, restated as a zero colour */
        static const float zero[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        const float *c = h->material < s->n_mat ? s->mats[h->material].color : zero;
        for (int k = 0; k < 4; ++k) rgba[k] = c[k] * ndl; /* alpha scaled too, may be negative */
    } else {
        rgba[0] = rgba[1] = rgba[2] = rgba[3] = 0.0f; /* KAT-6 */
    }
}

/* one pixel of the live path, plus the build-defined shadow ray:
 * origin = point + normal*1e-3, direction +Y (the direction the reference shades against,
 * wgsl:208); occluded => rgba *= 0.25. */
static uint32_t trace_pixel(const orc_scene *s, const orc_tree *tr, orc_walk_counts *cnt, const orc_view *v, uint32_t flags,
                            uint32_t px, uint32_t py, float rgba[4], uint32_t *prim, float *t) {
    float o[3], d[3];
    orc_hit h;
    uint32_t rays = 1;
    orc_ray_gen(v, px, py, o, d);
    if (tr) orc_hit_tree(s, tr, o, d, flags, 0, &h, cnt);
    else orc_hit_triangles(s, o, d, flags, &h);
    orc_shade(s, &h, rgba);
    if (h.hit && (flags & ORC_FLAG_SHADOW)) {
        float so[3], sd[3] = {0.0f, 1.0f, 0.0f};
        orc_hit sh;
        st3(so, add(ld3(h.point), scl(ld3(h.normal), ORC_SHADOW_EPS)));
        if (tr) orc_hit_tree(s, tr, so, sd, flags, 1, &sh, cnt);   /* hit_triangles(ray).hit: any accepted hit answers it */
        else orc_hit_triangles(s, so, sd, flags, &sh);
        rays = 2;
        if (sh.hit)
            for (int k = 0; k < 4; ++k) rgba[k] = rgba[k] * ORC_SHADOW_ATTEN;
    }
    *prim = h.hit ? h.prim : ORC_MISS_PRIM;
    *t = h.hit ? h.t : 0.0f;
    return rays;
}

typedef struct {
    const orc_scene *s; const orc_view *v; uint32_t flags;
    float *rgba; uint32_t *prim; float *t;
    const uint32_t *list; size_t n; int tid, nth;
    uint64_t rays;
    /* path */
    uint32_t spp, sample_first, max_segments, seed, stream; float *accum;
    uint32_t crop[4];   /* path: only pixels of this rectangle of the target (x, y, w, h; w == 0: no crop) */
    /* tree walk (NULL: the reference-shaped loop) */
    const orc_tree *tr; orc_walk_counts cnt;
} job_t;

static void view_rect(const orc_view *v, uint32_t *x0, uint32_t *y0, uint32_t *w, uint32_t *h) {
    if (v->vp_w) { *x0 = v->vp_x; *y0 = v->vp_y; *w = v->vp_w; *h = v->vp_h; }
    else { *x0 = 0; *y0 = 0; *w = v->width; *h = v->height; }
}

static void *primary_worker(void *arg) {
    job_t *j = (job_t *)arg;
    uint32_t W = j->v->width;
    uint32_t x0, y0, vw, vh;
    view_rect(j->v, &x0, &y0, &vw, &vh);
    j->rays = 0;
    memset(&j->cnt, 0, sizeof j->cnt);
    if (j->list) {
        for (size_t i = (size_t)j->tid; i < j->n; i += (size_t)j->nth) {
            uint32_t p = j->list[i], prim; float rgba[4], t;
            j->rays += trace_pixel(j->s, j->tr, &j->cnt, j->v, j->flags, p % W, p / W, rgba, &prim, &t);
            if (j->rgba) memcpy(j->rgba + 4 * i, rgba, sizeof rgba);
            if (j->prim) j->prim[i] = prim;
            if (j->t) j->t[i] = t;
        }
        return NULL;
    }
    for (uint32_t row = (uint32_t)j->tid; row < vh; row += (uint32_t)j->nth) {
        uint32_t py = y0 + row;
        for (uint32_t px = x0; px < x0 + vw; ++px) {
            size_t i = (size_t)py * W + px; uint32_t prim; float rgba[4], t;
            j->rays += trace_pixel(j->s, j->tr, &j->cnt, j->v, j->flags, px, py, rgba, &prim, &t);
            if (j->rgba) memcpy(j->rgba + 4 * i, rgba, sizeof rgba);
            if (j->prim) j->prim[i] = prim;
            if (j->t) j->t[i] = t;
        }
    }
    return NULL;
}

static uint64_t run_jobs(job_t *proto, int n_threads, void *(*fn)(void *)) {
    memset(&proto->cnt, 0, sizeof proto->cnt);
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 256) n_threads = 256;
    job_t *jobs = (job_t *)malloc(sizeof(job_t) * (size_t)n_threads);
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n_threads);
    uint64_t total = 0;
    for (int i = 0; i < n_threads; ++i) {
        jobs[i] = *proto; jobs[i].tid = i; jobs[i].nth = n_threads;
        if (i > 0) pthread_create(&th[i], NULL, fn, &jobs[i]);
    }
    fn(&jobs[0]);
    for (int i = 1; i < n_threads; ++i) pthread_join(th[i], NULL);
    for (int i = 0; i < n_threads; ++i) {
        total += jobs[i].rays;
        proto->cnt.node_visits += jobs[i].cnt.node_visits; proto->cnt.tri_tests += jobs[i].cnt.tri_tests;
        proto->cnt.mesh_culls += jobs[i].cnt.mesh_culls;
    }
    free(jobs); free(th);
    return total;
}

uint64_t orc_render_primary_tree(const orc_scene *s, const orc_tree *tr, const orc_view *v, uint32_t flags,
                                 float *rgba, uint32_t *prim, float *t,
                                 const uint32_t *pixel_list, size_t n_pixels, int n_threads, orc_walk_counts *cnt) {
    job_t j; memset(&j, 0, sizeof j);
    j.s = s; j.v = v; j.flags = flags; j.rgba = rgba; j.prim = prim; j.t = t;
    j.list = pixel_list; j.n = n_pixels; j.tr = tr;
    if (!pixel_list) { /* pixels outside the viewport are left cleared */
        size_t n = (size_t)v->width * v->height;
        if (rgba) memset(rgba, 0, n * 4 * sizeof(float));
        if (prim) memset(prim, 0xFF, n * sizeof(uint32_t));
        if (t) memset(t, 0, n * sizeof(float));
    }
    uint64_t rays = run_jobs(&j, n_threads, primary_worker);
    if (cnt) *cnt = j.cnt;
    return rays;
}
uint64_t orc_render_primary(const orc_scene *s, const orc_view *v, uint32_t flags,
                            float *rgba, uint32_t *prim, float *t,
                            const uint32_t *pixel_list, size_t n_pixels, int n_threads) {
    return orc_render_primary_tree(s, NULL, v, flags, rgba, prim, t, pixel_list, n_pixels, n_threads, NULL);
}

/* ---- A2 RNG: wgsl:38-74 --------------------------------------------------------------------- */
uint32_t orc_next_random(uint32_t *state) { /* wgsl:39-43 */
    *state = *state * 747796405u + 2891336453u;
    uint32_t st = *state;
    uint32_t word = ((st >> ((st >> 28u) + 4u)) ^ st) * 277803737u;
    return (word >> 22u) ^ word;
}
float orc_rand(uint32_t *state) { /* wgsl:53-55; 4294967295.0 rounds to 2^32 in f32 */
    return (float)orc_next_random(state) / 4294967295.0f;
}

/* The reference seeds with fract(sin(..)) (wgsl:44-47), which is not reproducible across
 * devices; deliberate deviation: integer seed = PCG-style hash of (pixel, sample, seed, stream). */
static uint32_t pcg_hash(uint32_t x) {
    uint32_t st = x * 747796405u + 2891336453u;
    uint32_t word = ((st >> ((st >> 28u) + 4u)) ^ st) * 277803737u;
    return (word >> 22u) ^ word;
}
uint32_t orc_seed(uint32_t pixel, uint32_t sample, uint32_t seed, uint32_t stream) {
    uint32_t h = pcg_hash(pixel);
    h = pcg_hash(h ^ (sample * 0x9E3779B9u + 0x7F4A7C15u));
    h = pcg_hash(h ^ (seed + 0x85EBCA6Bu));
    h = pcg_hash(h ^ (stream * 0xC2B2AE35u + 0x27D4EB2Fu));
    return h;
}

/* Deterministic log/cos from +,-,*,/ and integer bit moves only, so CPU and GPU agree bit for
 * bit (the WGSL log/cos are implementation-defined in accuracy anyway). */
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

float orc_det_logf(float x) {
    /* x > 0 finite normal assumed for accuracy; x == 0 -> -inf, x < 0 or NaN -> NaN */
    if (x != x || x < 0.0f) return u2f(0x7FC00000u);
    if (x == 0.0f) return u2f(0xFF800000u);
    uint32_t ix = f2u(x);
    int k = 0;
    if (ix < 0x00800000u) { x = x * 8388608.0f; ix = f2u(x); k = -23; } /* subnormal */
    if (ix >= 0x7F800000u) return x; /* +inf */
    /* reduce to m in [sqrt(2)/2, sqrt(2)) */
    ix += 0x3F800000u - 0x3F3504F3u;
    k += (int)(ix >> 23) - 127;
    ix = (ix & 0x007FFFFFu) + 0x3F3504F3u;
    float m = u2f(ix);
    float f = m - 1.0f;
    float s = f / (2.0f + f);
    float z = s * s;
    float w = z * z;
    /* fdlibm-style minimax coefficients for log(1+f) */
    float t1 = w * (0.40000972152f + w * 0.24279078841f);
    float t2 = z * (0.66666662693f + w * 0.28498786688f);
    float R = t2 + t1;
    float hfsq = 0.5f * f * f;
    float dk = (float)k;
    return dk * 6.9313812256e-01f + ((s * (hfsq + R) + dk * 9.0580006145e-06f) - hfsq + f);
}

float orc_det_cosf(float x) {
    /* accurate for |x| up to a few thousand; used on theta in [0, 2*pi] */
    if (x != x || x - x != 0.0f) return u2f(0x7FC00000u);
    float ax = x < 0.0f ? -x : x;
    float q = floorf(ax * 0.63661977237f + 0.5f); /* nearest multiple of pi/2 */
    /* Cody-Waite: pi/2 = 1.5703125 + 4.837512969970703125e-4 + 7.54978995489188e-8 */
    float r = ((ax - q * 1.5703125f) - q * 4.837512969970703125e-4f) - q * 7.54978995489188e-8f;
    int n = (int)q & 3;
    float r2 = r * r;
    float sp = r + r * r2 * (-1.6666654611e-1f + r2 * (8.3321608736e-3f + r2 * -1.9515295891e-4f));
    float cp = (1.0f - 0.5f * r2) + r2 * r2 * (4.166664568298827e-2f + r2 * (-1.388731625493765e-3f + r2 * 2.443315711809948e-5f));
    switch (n) {
        case 0: return cp;
        case 1: return -sp;
        case 2: return -cp;
        default: return sp;
    }
}

float orc_rand_norm(uint32_t *state) { /* wgsl:57-61; theta's rand() is drawn first */
    float theta = 2.0f * 3.1415926f * orc_rand(state);
    float rho = sqrtf(-2.0f * orc_det_logf(orc_rand(state)));
    return rho * orc_det_cosf(theta);
}
void orc_rand_dir(uint32_t *state, float d[3]) { /* wgsl:63-65; x, y, z drawn in order */
    float x = orc_rand_norm(state);
    float y = orc_rand_norm(state);
    float z = orc_rand_norm(state);
    st3(d, normalize(V(x, y, z)));
}
void orc_rand_hemi(uint32_t *state, const float n[3], float d[3]) { /* wgsl:67-74 */
    orc_rand_dir(state, d);
    if (dot(ld3(n), ld3(d)) < 0.0f) { d[0] = -d[0]; d[1] = -d[1]; d[2] = -d[2]; }
}

/* ---- A7/A8 progressive path tracing: loop shape of wgsl:180-202 ------------------------------ */
static void *path_worker(void *arg) {
    job_t *j = (job_t *)arg;
    const orc_scene *s = j->s;
    uint32_t W = j->v->width;
    uint32_t x0, y0, vw, vh;
    view_rect(j->v, &x0, &y0, &vw, &vh);
    if (j->crop[2]) {   /* the covered rectangle, cut down to the crop: the same pixels get the same values, fewer of them */
        uint32_t cx0 = j->crop[0] > x0 ? j->crop[0] : x0, cy0 = j->crop[1] > y0 ? j->crop[1] : y0;
        uint64_t cx1 = (uint64_t)j->crop[0] + j->crop[2], cy1 = (uint64_t)j->crop[1] + j->crop[3];
        if (cx1 > (uint64_t)x0 + vw) cx1 = (uint64_t)x0 + vw;
        if (cy1 > (uint64_t)y0 + vh) cy1 = (uint64_t)y0 + vh;
        vw = cx1 > cx0 ? (uint32_t)(cx1 - cx0) : 0; vh = cy1 > cy0 ? (uint32_t)(cy1 - cy0) : 0;
        x0 = cx0; y0 = cy0;
    }
    j->rays = 0;
    memset(&j->cnt, 0, sizeof j->cnt);
    for (uint32_t row = (uint32_t)j->tid; row < vh; row += (uint32_t)j->nth) {
        uint32_t py = y0 + row;
        for (uint32_t px = x0; px < x0 + vw; ++px) {
            uint32_t pixel = py * W + px;
            float o[3], d[3];
            orc_ray_gen(j->v, px, py, o, d);
            float *acc = j->accum + 4 * (size_t)pixel;
            for (uint32_t k = 0; k < j->spp; ++k) {
                uint32_t state = orc_seed(pixel, j->sample_first + k, j->seed, j->stream);
                v3 col = V(1.0f, 1.0f, 1.0f);
                v3 light = V(0.0f, 0.0f, 0.0f);
                /* Ray(origin, dir + vec3(rand(), rand(), 0)*0.0015), not renormalised (wgsl:184) */
                float jx = orc_rand(&state), jy = orc_rand(&state);
                float ro[3] = {o[0], o[1], o[2]};
                float rd[3];
                st3(rd, add(ld3(d), scl(V(jx, jy, 0.0f), 0.0015f)));
                for (uint32_t i = 0; i < j->max_segments; ++i) {
                    orc_hit rec;
                    if (j->tr) orc_hit_tree(s, j->tr, ro, rd, j->flags, 0, &rec, &j->cnt);
                    else orc_hit_triangles(s, ro, rd, j->flags, &rec);
                    j->rays++;
                    if (rec.hit) {
                        /* record.color = material colour rgb; record.light = 0 (T4 has no emission) */
                        float nd[3];
                        orc_rand_hemi(&state, rec.normal, nd);
                        memcpy(ro, rec.point, sizeof ro);
                        memcpy(rd, nd, sizeof rd);
                        col = rec.material < s->n_mat ? mul(col, ld3(s->mats[rec.material].color)) : V(0.0f, 0.0f, 0.0f);
                    } else {
                        light = add(light, mul(V(0.5f, 0.71f, 0.86f), col)); /* sky, wgsl:197 */
                        break;
                    }
                }
                acc[0] = acc[0] + light.x; acc[1] = acc[1] + light.y; acc[2] = acc[2] + light.z;
                acc[3] = acc[3] + 1.0f;
            }
        }
    }
    return NULL;
}

uint64_t orc_render_path_tree(const orc_scene *s, const orc_tree *tr, const orc_view *v, uint32_t flags,
                              uint32_t spp, uint32_t sample_first, uint32_t max_segments,
                              uint32_t seed, uint32_t stream, float *accum, int n_threads, orc_walk_counts *cnt) {
    job_t j; memset(&j, 0, sizeof j);
    j.s = s; j.v = v; j.flags = flags; j.spp = spp; j.sample_first = sample_first;
    j.max_segments = max_segments; j.seed = seed; j.stream = stream; j.accum = accum; j.tr = tr;
    uint64_t rays = run_jobs(&j, n_threads, path_worker);
    if (cnt) *cnt = j.cnt;
    return rays;
}
uint64_t orc_render_path_tree_crop(const orc_scene *s, const orc_tree *tr, const orc_view *v, uint32_t flags,
                                   uint32_t spp, uint32_t sample_first, uint32_t max_segments,
                                   uint32_t seed, uint32_t stream, const uint32_t crop[4], float *accum, int n_threads, orc_walk_counts *cnt) {
    job_t j; memset(&j, 0, sizeof j);
    j.s = s; j.v = v; j.flags = flags; j.spp = spp; j.sample_first = sample_first;
    j.max_segments = max_segments; j.seed = seed; j.stream = stream; j.accum = accum; j.tr = tr;
    if (crop) memcpy(j.crop, crop, sizeof j.crop);
    uint64_t rays = run_jobs(&j, n_threads, path_worker);
    if (cnt) *cnt = j.cnt;
    return rays;
}
uint64_t orc_render_path(const orc_scene *s, const orc_view *v, uint32_t flags,
                         uint32_t spp, uint32_t sample_first, uint32_t max_segments,
                         uint32_t seed, uint32_t stream, float *accum, int n_threads) {
    return orc_render_path_tree(s, NULL, v, flags, spp, sample_first, max_segments, seed, stream, accum, n_threads, NULL);
}

void orc_resolve(const float *accum, size_t n_px, uint32_t total_samples, float *rgba) {
    float n = (float)total_samples;
    for (size_t i = 0; i < n_px * 4; ++i) rgba[i] = accum[i] / n; /* light / f32(samples), wgsl:212 */
}

/* ---- presentation: Bgra8UnormSrgb target, src/pipeline.rs:56-60 ------------------------------- */
static uint8_t unorm8(float x) {
    if (!(x > 0.0f)) x = 0.0f; /* NaN -> 0 */
    if (x > 1.0f) x = 1.0f;
    return (uint8_t)floorf(x * 255.0f + 0.5f);
}
/* Colour channels: byte = round(255 * srgb_oetf(x)), as the number of linear-space thresholds <= x.  srgb_table.h
 * (tools/gen_srgb_table.py) holds, for k = 1..255, the smallest binary32 x whose byte is >= k, exactly rounded from
 * the OETF -- float compares only, so this function has one answer on every machine (a pow() here and a powf() on
 * the device differed by 1 LSB on ~1 % of bytes).  NaN and x <= 0 give 0, x >= 1 gives 255.
 * tests/test_srgb_table.py pins the table to the formula evaluated in binary64. */
#include "srgb_table.h"
static uint8_t srgb_byte(float x) {
    uint32_t n = 0;
    for (uint32_t step = 128; step; step >>= 1) {
        union { uint32_t u; float f; } t;
        t.u = RT_SRGB_THRESHOLD_BITS[n + step - 1];
        if (t.f <= x) n += step;
    }
    return (uint8_t)n;
}
void orc_encode_bgra8_srgb(const float *rgba, size_t n_px, uint8_t *bgra) {
    for (size_t i = 0; i < n_px; ++i) {
        bgra[4 * i + 0] = srgb_byte(rgba[4 * i + 2]);
        bgra[4 * i + 1] = srgb_byte(rgba[4 * i + 1]);
        bgra[4 * i + 2] = srgb_byte(rgba[4 * i + 0]);
        bgra[4 * i + 3] = unorm8(rgba[4 * i + 3]); /* alpha is linear */
    }
}

/* ---- P1 prepare_meshinfo: src/ray_tracing.rs:98-169 --------------------------------------------- */
/* glam Mat3::from_quat and Quat*Vec3 restated (un-vendored dependency glam 0.27.0, Cargo.lock:2129;
 * parity unpinned). */
static void mat3_from_quat(const float q[4], v3 *c0, v3 *c1, v3 *c2) {
    float x = q[0], y = q[1], z = q[2], w = q[3];
    float x2 = x + x, y2 = y + y, z2 = z + z;
    float xx = x * x2, xy = x * y2, xz = x * z2, yy = y * y2, yz = y * z2, zz = z * z2;
    float wx = w * x2, wy = w * y2, wz = w * z2;
    *c0 = V(1.0f - (yy + zz), xy + wz, xz - wy);
    *c1 = V(xy - wz, 1.0f - (xx + zz), yz + wx);
    *c2 = V(xz + wy, yz - wx, 1.0f - (xx + yy));
}
static v3 quat_mul_vec3(const float q[4], v3 r) {
    float w = q[3];
    v3 b = V(q[0], q[1], q[2]);
    float b2 = dot(b, b);
    return add(add(scl(r, w * w - b2), scl(b, dot(r, b) * 2.0f)), scl(cross(b, r), w * 2.0f));
}
static v3 mat3_mul(v3 c0, v3 c1, v3 c2, v3 p) { return add(add(scl(c0, p.x), scl(c1, p.y)), scl(c2, p.z)); }

void orc_prepare_meshinfo(const orc_entity *ents, size_t n_ent,
                          orc_triangle *tris, size_t *n_tri,
                          orc_mesh_info *meshes, size_t *n_mesh,
                          orc_vertex *verts, size_t *n_vtx,
                          orc_material *mats, size_t *n_mat) {
    size_t nt = 0, nm = 0, nv = 0, nc = 0;
    uint32_t material_index = 0;
    for (size_t e = 0; e < n_ent; ++e) {
        const orc_entity *E = &ents[e];
        memcpy(mats[nc++].color, E->base_color_linear, 16); /* pushed BEFORE the attribute check (:122) */
        if (!E->positions || !E->normals) continue;          /* "Mesh missing attribute" (:123-129) */
        size_t vertices_len = nv;
        v3 c0, c1, c2;
        mat3_from_quat(E->rotation, &c0, &c1, &c2);
        /* affine matrix3 = R * diag(scale) */
        v3 m0 = scl(c0, E->scale[0]), m1 = scl(c1, E->scale[1]), m2 = scl(c2, E->scale[2]);
        for (uint32_t i = 0; i < E->n_vtx; ++i) {
            orc_vertex *o = &verts[nv++];
            memset(o, 0, sizeof *o);
            /* transform_point (:137); normals rotate only, no scale, no renormalise (:139-141) */
            st3(o->pos, add(mat3_mul(m0, m1, m2, ld3(E->positions + 3 * i)), ld3(E->translation)));
            st3(o->norm, quat_mul_vec3(E->rotation, ld3(E->normals + 3 * i)));
        }
        /* Cuboid::from_size(half_extents*2).aabb_3d(translate, rotate) (:148-149): ignores the
         * Aabb centre and the scale; half = |R| * half_extents, centred on the translation. */
        v3 he = V((E->aabb_half_extents[0] * 2.0f) / 2.0f, (E->aabb_half_extents[1] * 2.0f) / 2.0f,
                  (E->aabb_half_extents[2] * 2.0f) / 2.0f);
        v3 a0 = V(fabsf(c0.x), fabsf(c0.y), fabsf(c0.z));
        v3 a1 = V(fabsf(c1.x), fabsf(c1.y), fabsf(c1.z));
        v3 a2 = V(fabsf(c2.x), fabsf(c2.y), fabsf(c2.z));
        v3 half = mat3_mul(a0, a1, a2, he);
        orc_mesh_info *M = &meshes[nm++];
        memset(M, 0, sizeof *M);
        M->first_tri = (uint32_t)nt;
        M->tri_count = E->n_idx / 3u;
        M->material = material_index;
        st3(M->aabb_min, sub(ld3(E->translation), half));
        st3(M->aabb_max, add(ld3(E->translation), half));
        material_index += 1;
        for (uint32_t i = 0; i + 2 < E->n_idx; i += 3) { /* tuples() drops a trailing partial triple (:161) */
            orc_triangle *T = &tris[nt++];
            T->idx[0] = (uint32_t)(E->indices[i + 0] + vertices_len);
            T->idx[1] = (uint32_t)(E->indices[i + 1] + vertices_len);
            T->idx[2] = (uint32_t)(E->indices[i + 2] + vertices_len);
        }
    }
    *n_tri = nt; *n_mesh = nm; *n_vtx = nv; *n_mat = nc;
}
