// rt_tools.hip -- measurement aid behind bench.py's roofline (no counterpart in the reference, not on the render path):
// what the memory system of THIS GPU delivers to the access pattern of a BVH walk.  Every lane reads one 64-B record
// (4 x dwordx4, as a node step does) at its own pseudo-random 64-B aligned address; the buffer's size decides which level of
// the hierarchy serves the gathers (16 KiB: vector L1, 2 MiB: every XCD's L2, 128 MiB: Infinity Cache, 1 GiB / 4 GiB: HBM),
// at the traversal kernels' occupancy (8 blocks of 256 threads per CU), in two forms:
//   independent : a lane's next address does not depend on what it loaded (the unrolled loop keeps several gathers in flight
//                 per lane): the THROUGHPUT ceiling of 64-B gathers from that level
//   dependent   : the next address is computed from the loaded record (a pointer chase, one gather in flight per lane -- what
//                 a ray's walk is): lanes in flight / latency, the latency-bound rate at this occupancy
// bench.py's algorithmic-bytes figure (node visits x 64 B / time) is a gather rate in these units; "8 TB/s HBM" is not its
// ceiling.  Runs in the caller's process: no child process, no exec.
#include <hip/hip_runtime.h>
#include "../../include/rt_amd.h"

namespace {
__device__ __forceinline__ unsigned hash32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
template <bool DEP>
__global__ __launch_bounds__(256, 8) void k_tool_gather(const uint4 *__restrict__ buf, unsigned mask, int iters, unsigned *out) {
    unsigned idx = hash32(blockIdx.x * 256u + threadIdx.x + 1u);
    unsigned acc = 0;
    if (DEP) {
        for (int i = 0; i < iters; ++i) {
            const uint4 *p = buf + (size_t)(idx & mask) * 4u;
            uint4 a = p[0], b = p[1], c = p[2], d = p[3];
            idx = hash32(idx ^ a.x ^ b.y ^ c.z ^ d.w);     // the next address needs all four loads, as a node step does
        }
        acc = idx;
    } else {
#pragma unroll 4
        for (int i = 0; i < iters; ++i) {
            const uint4 *p = buf + (size_t)(idx & mask) * 4u;
            uint4 a = p[0], b = p[1], c = p[2], d = p[3];
            acc += a.x ^ b.y ^ c.z ^ d.w;
            idx = hash32(idx + 0x9E3779B9u);
        }
    }
    if (acc == 0x12345678u) out[0] = acc;       // keeps the loads alive
}
__global__ void k_tool_fill(uint4 *buf, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) buf[i] = make_uint4(hash32((unsigned)i), hash32((unsigned)i + 7u), hash32((unsigned)i + 13u), hash32((unsigned)i + 29u));
}
}  // namespace

extern "C" int rt_measure_gather_rates(int device_ordinal, double *ggather_per_s, size_t n_values) {
    if (!ggather_per_s || n_values < RT_GATHER_RATE_VALUES) return RT_ERR_ARG;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0) return RT_ERR_NO_GPU;
    if (device_ordinal < 0 || device_ordinal >= n_dev) return RT_ERR_ARG;
    hipDeviceProp_t prop;
    int prev_dev = -1;
    (void)hipGetDevice(&prev_dev);                 // the caller's current device is put back before returning
    if (hipSetDevice(device_ordinal) != hipSuccess || hipGetDeviceProperties(&prop, device_ordinal) != hipSuccess) {
        if (prev_dev >= 0) (void)hipSetDevice(prev_dev);
        return RT_ERR_HIP;
    }
    const size_t sizes[RT_GATHER_RATE_VALUES / 2] = {(size_t)16 << 10, (size_t)2 << 20, (size_t)128 << 20, (size_t)1 << 30, (size_t)4 << 30};
    const size_t max_bytes = sizes[RT_GATHER_RATE_VALUES / 2 - 1];
    uint4 *buf = nullptr; unsigned *out = nullptr;
    hipStream_t st = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = RT_OK;
    do {
        if (hipMalloc((void **)&buf, max_bytes) != hipSuccess) { (void)hipGetLastError(); rc = RT_ERR_OOM; break; }
        if (hipMalloc((void **)&out, 4) != hipSuccess || hipStreamCreate(&st) != hipSuccess ||
            hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { rc = RT_ERR_HIP; break; }
        hipLaunchKernelGGL(k_tool_fill, dim3(4096), dim3(256), 0, st, buf, max_bytes / 16);
        const int grid = prop.multiProcessorCount * 8;
        for (int c = 0; c < RT_GATHER_RATE_VALUES / 2 && rc == RT_OK; ++c) {
            const unsigned mask = (unsigned)(sizes[c] / 64 - 1);
            for (int dep = 0; dep < 2; ++dep) {
                const int iters = dep ? 256 : 512;
                for (int rep = 0; rep < 2; ++rep) {      // the first run warms the caches
                    (void)hipEventRecord(e0, st);
                    if (dep) hipLaunchKernelGGL(k_tool_gather<true>, dim3(grid), dim3(256), 0, st, buf, mask, iters, out);
                    else hipLaunchKernelGGL(k_tool_gather<false>, dim3(grid), dim3(256), 0, st, buf, mask, iters, out);
                    (void)hipEventRecord(e1, st);
                    if (hipGetLastError() != hipSuccess || hipEventSynchronize(e1) != hipSuccess) { rc = RT_ERR_HIP; break; }
                }
                float ms = 0.0f;
                if (rc != RT_OK || hipEventElapsedTime(&ms, e0, e1) != hipSuccess || !(ms > 0.0f)) { rc = RT_ERR_HIP; break; }
                ggather_per_s[2 * c + dep] = (double)grid * 256.0 * iters / (ms * 1e-3) / 1e9;
            }
        }
    } while (0);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (st) (void)hipStreamDestroy(st);
    if (out) (void)hipFree(out);
    if (buf) (void)hipFree(buf);
    if (prev_dev >= 0 && prev_dev != device_ordinal) (void)hipSetDevice(prev_dev);
    return rc;
}
