// Scratch micro-benchmark (not part of the product): what read bandwidth does gfx950 give for different
// ways of assigning contiguous chunks to waves?  hipcc --offload-arch=gfx950 -O3 tools/probe_stream.hip -o /tmp/probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
// every wave reads `per_wave` bytes as 1-KiB wave-loads (16 B/lane); chunk c of wave w lives at
//   mode 0 (grid-stride)   : (c * total_waves + w) * chunk
//   mode 1 (wave-owned)    : (w * chunks_per_wave + c) * chunk
//   mode 2 (wg-owned)      : ((wg * chunks_per_wave + c) * waves_per_wg + wave_in_wg) * chunk
template <int UNR>
__global__ void __launch_bounds__(256) probe(const uint4* __restrict__ buf, unsigned* out, long chunk_vec, long chunks_per_wave,
                                             int mode, long total_waves) {
    const int lane = threadIdx.x & 63;
    const long w = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int wpw = blockDim.x >> 6, wiw = threadIdx.x >> 6;
    unsigned acc = 0;
    for (long c = 0; c < chunks_per_wave; ++c) {
        long base;
        if (mode == 0) base = (c * total_waves + w) * chunk_vec;
        else if (mode == 1) base = (w * chunks_per_wave + c) * chunk_vec;
        else base = (((long)blockIdx.x * chunks_per_wave + c) * wpw + wiw) * chunk_vec;
        for (long i = lane; i < chunk_vec; i += 64 * UNR) {
            uint4 v[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) v[u] = (i + 64 * u < chunk_vec) ? buf[base + i + 64 * u] : make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int u = 0; u < UNR; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}
int main(int argc, char** argv) {
    const size_t bytes = 1024ull << 20;
    uint4* buf; unsigned* out;
    hipMalloc(&buf, bytes); hipMalloc(&out, 4); hipMemset(buf, 1, bytes);
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    const int wpws[] = {4, 8};
    for (int wi = 0; wi < 2; ++wi)
    for (long chunk : {1024L, 3072L, 16384L, 65536L})
    for (int mode = 0; mode < 3; ++mode)
    for (long nwg : {512L, 1024L, 2048L, 8192L}) {
        const int wpw = wpws[wi];
        const long total_waves = nwg * wpw;
        const long cpw = (long)(bytes / chunk / total_waves);
        if (cpw < 1) continue;
        const size_t used = (size_t)cpw * total_waves * chunk;
        std::vector<float> t;
        for (int it = 0; it < 7; ++it) {
            hipEventRecord(s);
            hipLaunchKernelGGL(probe<4>, dim3(nwg), dim3(64 * wpw), 0, 0, buf, out, chunk / 16, cpw, mode, total_waves);
            hipEventRecord(e); hipEventSynchronize(e);
            float ms; hipEventElapsedTime(&ms, s, e); t.push_back(ms);
        }
        std::sort(t.begin(), t.end());
        printf("wpw=%d chunk=%6ld mode=%d nwg=%5ld cpw=%5ld : %7.1f us  %6.0f GB/s\n", wpw, chunk, mode, nwg, cpw, t[3] * 1e3, used / t[3] / 1e6);
    }
    return 0;
}
