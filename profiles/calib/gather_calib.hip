// gather_calib.hip -- calibrates rocprofv3's FETCH_SIZE for THIS kernel family's access pattern:
// every lane reads one 64-B record (4 x dwordx4) at a pseudo-random, 64-B aligned address of a buffer
// far larger than the 256 MiB Infinity Cache, each record touched once => known unique bytes.
// MI355X_MICROARCH.md (HBM): FETCH_SIZE reads 1/2 of a wide coalesced stream on gfx950 and is
// uncalibrated for other widths; this gives the factor for 64-B gathers.
// Build: hipcc --offload-arch=gfx950 -O3 -o gather_calib gather_calib.hip
// Run:   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -- ./gather_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void k_gather64(const float4 *__restrict__ buf, size_t n_rec, float *out, unsigned long long mult) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t r = (i * mult) % n_rec;              // mult odd and coprime with n_rec (power of two): a permutation
    const float4 *p = buf + r * 4;
    float4 a = p[0], b = p[1], c = p[2], d = p[3];
    float s = a.x + b.y + c.z + d.w;
    if (s == 12345.678f) out[0] = s;           // keep the loads alive
}
__global__ void k_stream16(const float4 *__restrict__ buf, size_t n, float *out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { float4 a = buf[i]; if (a.x == 12345.678f) out[0] = a.y; }
}
int main() {
    const size_t n_rec = (size_t)1 << 26;        // 64 Mi records x 64 B = 4 GiB
    float4 *buf; float *out;
    if (hipMalloc(&buf, n_rec * 64) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 0, n_rec * 64);
    const size_t lanes = (size_t)1 << 24;        // 16 Mi lanes -> 1 GiB of unique records
    hipLaunchKernelGGL(k_gather64, dim3(lanes / 256), dim3(256), 0, 0, buf, n_rec, out, 0x9E3779B1ull);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(k_stream16, dim3((n_rec * 4 / 4) / 256), dim3(256), 0, 0, buf, n_rec, out);   // 1 GiB stream
    hipDeviceSynchronize();
    printf("k_gather64: %zu unique bytes; k_stream16: %zu bytes\n", lanes * 64, n_rec * 16);
    return 0;
}
