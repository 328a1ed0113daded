// rt_wide.h -- quantisation of a wide node's child boxes, shared by the host builder (g++) and the
// device re-quantisation after a refit (hipcc): plain IEEE f32 operations in a fixed order, built
// with -ffp-contract=off on both sides, so host and device produce the same bytes.
#pragma once
#include <math.h>
#include "rt_layout.h"

#if defined(__HIPCC__)
#define RT_HD __host__ __device__
#else
#define RT_HD
#endif

// Fills org / scl / lo* / hi* of W from the binary-tree boxes named by src (child codes are left alone).
// Containment: every source box is first expanded by M = 2e-6 * max|coordinate| + 1e-12 per axis -- more
// than the rounding of this function (a few ulp) plus that of the traversal's  byte*(scl*inv) - (o-org)*inv
// form relative to the node's own size, and enough for the +-1e30 that form uses for 1/0 -- and then rounded
// outward to the grid.  An axis whose extent is
// not finite gets scl = +inf: the traversal then computes NaN / inf planes, which its min/max drop, i.e.
// "no constraint on this axis" (conservative).
RT_HD inline void rt_wide_quantise(const RtBvhNode *nodes, const RtWideSrc &src, RtWideNode &W) {
    const float inf = __builtin_huge_valf();
    float lo[4][3], hi[4][3];
    float nlo[3] = {inf, inf, inf}, nhi[3] = {-inf, -inf, -inf};
    for (int k = 0; k < 4; ++k) {
        if (src.slot[k] == RT_WIDE_EMPTY) continue;
        const RtBvhNode &N = nodes[src.slot[k] >> 1];
        const float *l = (src.slot[k] & 1u) ? N.lo1 : N.lo0, *h = (src.slot[k] & 1u) ? N.hi1 : N.hi0;
        for (int a = 0; a < 3; ++a) {
            lo[k][a] = l[a]; hi[k][a] = h[a];
            nlo[a] = fminf(nlo[a], l[a]); nhi[a] = fmaxf(nhi[a], h[a]);
        }
    }
    uint32_t qlo[3] = {0, 0, 0}, qhi[3] = {0, 0, 0};
    float org[3], scl[3];
    for (int a = 0; a < 3; ++a) {
        const float amax = fmaxf(fabsf(nlo[a]), fabsf(nhi[a]));
        const float M = 2e-6f * amax + 1e-12f;
        float o = nlo[a] - M;
        float s = ((nhi[a] - nlo[a]) + 2.0f * M) / 253.0f;
        const bool bounded = s < inf && s > 0.0f && fabsf(o) < inf;     // false for an empty node, too
        if (!bounded) { o = 0.0f; s = inf; }
        org[a] = o; scl[a] = s;
        for (int k = 0; k < 4; ++k) {
            uint32_t bl = 255u, bh = 0u;                                // empty slot: inverted box
            if (src.slot[k] != RT_WIDE_EMPTY) {
                bl = 0u; bh = 255u;
                if (bounded) {
                    float fl = floorf(((lo[k][a] - M) - o) / s);
                    float fh = ceilf(((hi[k][a] + M) - o) / s);
                    fl = fl < 0.0f ? 0.0f : (fl > 255.0f ? 255.0f : fl);
                    fh = fh < 0.0f ? 0.0f : (fh > 255.0f ? 255.0f : fh);
                    bl = (uint32_t)fl; bh = (uint32_t)fh;
                    // belt and braces: the grid plane, evaluated in f32, must not lie inside the source box
                    if (bl > 0u && o + (float)bl * s > lo[k][a]) --bl;
                    if (bh < 255u && o + (float)bh * s < hi[k][a]) ++bh;
                    if (o + (float)bh * s < hi[k][a]) { bl = 0u; bh = 255u; org[a] = 0.0f; scl[a] = inf; }
                }
            }
            qlo[a] |= bl << (8 * k); qhi[a] |= bh << (8 * k);
        }
    }
    // an axis that fell back to "unbounded" after some children were packed: every real child gets 0 / 255 there
    for (int a = 0; a < 3; ++a)
        if (!(scl[a] < inf)) {
            qlo[a] = 0u; qhi[a] = 0u;
            for (int k = 0; k < 4; ++k) {
                const bool used = src.slot[k] != RT_WIDE_EMPTY;
                qlo[a] |= (used ? 0u : 255u) << (8 * k); qhi[a] |= (used ? 255u : 0u) << (8 * k);
            }
        }
    W.org[0] = org[0]; W.org[1] = org[1]; W.org[2] = org[2];
    W.sclx = scl[0]; W.scly = scl[1]; W.sclz = scl[2];
    W.lox = qlo[0]; W.loy = qlo[1]; W.loz = qlo[2];
    W.hix = qhi[0]; W.hiy = qhi[1]; W.hiz = qhi[2];
}
