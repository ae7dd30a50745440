// Scratch micro-benchmark (not part of the product): the floor of a weight-streaming launch at decode sizes.
// 32 launches back to back, each reading its own region of `bytes` once with nt 16-B loads and doing nothing else.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_small.hip -o tools/probe_small.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// each wave reads `n` consecutive 1-KiB lines, DEPTH lines in flight
template <int DEPTH>
__global__ void __launch_bounds__(256) stream_kernel(const u32x4* __restrict__ buf, unsigned* out, int n) {
    const int lane = threadIdx.x & 63;
    const long w = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const u32x4* p = buf + w * (long)n * 64 + lane;
    unsigned acc = 0;
    for (int i = 0; i < n; i += DEPTH) {
        u32x4 v[DEPTH];
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) v[u] = __builtin_nontemporal_load(p + (long)(i + u < n ? i + u : n - 1) * 64);
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) acc ^= v[u].x ^ v[u].w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
int main() {
    const size_t region = 72ull << 20, total = region * 32;
    u32x4* buf; unsigned* out;
    if (hipMalloc(&buf, total) != hipSuccess) return 1;
    hipMalloc(&out, 4); hipMemset(buf, 1, total);
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    const double sizes_mb[] = {9.4, 14.7, 40.6, 66.0};
    for (double mb : sizes_mb)
    for (int lines_per_wave : {2, 4, 9, 18, 36})
    for (int depth : {2, 4, 9}) {
        if (depth > lines_per_wave) continue;
        const long lines = (long)(mb * 1e6 / 1024);
        const long waves = lines / lines_per_wave, wgs = waves / 4;
        std::vector<float> t;
        for (int it = 0; it < 5; ++it) {
            hipEventRecord(s, st);
            for (int l = 0; l < 32; ++l) {
                const u32x4* p = buf + (region / 16) * l;
                if (depth == 2) hipLaunchKernelGGL(stream_kernel<2>, dim3(wgs), dim3(256), 0, st, p, out, lines_per_wave);
                else if (depth == 4) hipLaunchKernelGGL(stream_kernel<4>, dim3(wgs), dim3(256), 0, st, p, out, lines_per_wave);
                else hipLaunchKernelGGL(stream_kernel<9>, dim3(wgs), dim3(256), 0, st, p, out, lines_per_wave);
            }
            hipEventRecord(e, st); hipEventSynchronize(e);
            float ms; hipEventElapsedTime(&ms, s, e); t.push_back(ms / 32);
        }
        std::sort(t.begin(), t.end());
        printf("%5.1f MB lines/wave=%2d depth=%d wgs=%5ld : %6.2f us  %5.0f GB/s\n", mb, lines_per_wave, depth, wgs, t[2] * 1e3,
               wgs * 4 * lines_per_wave * 1024.0 / t[2] / 1e6);
    }
    return 0;
}
