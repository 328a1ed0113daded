// gather_variants.hip -- experiment (round 3): what paces the CUs' vector-memory front end for cache-resident gathers?
// Same kernel shape as csrc/rt_tools.hip (8 blocks of 256 threads per CU, random 64-B aligned records of a 2 MiB buffer that
// every XCD's L2 holds), varied in (a) LOADS per record: 1..4 dwordx4 per lane, and (b) SHARE: groups of 1, 4, 16 or 64
// consecutive lanes of a wave read the SAME record (lanes of a wave that walk the top of a tree hold the same node).
// Build: hipcc --offload-arch=gfx950 -O3 -o gather_variants gather_variants.hip      Run: ./gather_variants
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ unsigned hash32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
template <int LOADS>
__global__ __launch_bounds__(256, 8) void k(const uint4 *__restrict__ buf, unsigned mask, int iters, unsigned share_mask, unsigned *out) {
    unsigned idx = hash32((blockIdx.x * 256u + threadIdx.x) & ~share_mask);
    unsigned acc = 0;
#pragma unroll 4
    for (int i = 0; i < iters; ++i) {
        const uint4 *p = buf + (size_t)(idx & mask) * 4u;
        uint4 a = p[0]; acc += a.x;
        if (LOADS > 1) { uint4 b = p[1]; acc += b.y; }
        if (LOADS > 2) { uint4 c = p[2]; acc += c.z; }
        if (LOADS > 3) { uint4 d = p[3]; acc += d.w; }
        idx = hash32(idx + 0x9E3779B9u);
    }
    if (acc == 0x12345678u) out[0] = acc;
}
int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const size_t bytes = (size_t)2 << 20;
    uint4 *buf; unsigned *out;
    hipMalloc(&buf, bytes); hipMalloc(&out, 4); hipMemset(buf, 1, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = prop.multiProcessorCount * 8, iters = 512;
    printf("%d CUs; 2 MiB buffer (L2-resident); G records/s and G lane-loads (dwordx4)/s\n", prop.multiProcessorCount);
    printf("%-8s %-8s %14s %16s\n", "loads", "share", "Grecords/s", "Glane-loads/s");
    for (int share = 1; share <= 64; share *= 4)
        for (int loads = 1; loads <= 4; ++loads) {
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0, 0);
                const unsigned m = (unsigned)(bytes / 64 - 1), sm = (unsigned)share - 1u;
                if (loads == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, buf, m, iters, sm, out);
                if (loads == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, buf, m, iters, sm, out);
                if (loads == 3) hipLaunchKernelGGL(k<3>, dim3(grid), dim3(256), 0, 0, buf, m, iters, sm, out);
                if (loads == 4) hipLaunchKernelGGL(k<4>, dim3(grid), dim3(256), 0, 0, buf, m, iters, sm, out);
                hipEventRecord(e1, 0); hipEventSynchronize(e1);
            }
            hipEventElapsedTime(&ms, e0, e1);
            const double recs = (double)grid * 256.0 * iters / (ms * 1e-3) / 1e9;
            printf("%-8d %-8d %14.1f %16.1f\n", loads, share, recs, recs * loads);
        }
    return 0;
}
