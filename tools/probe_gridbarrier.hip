// Probe: cost of a software grid barrier on MI355X for a persistent kernel (all workgroups co-resident).
// hipcc --offload-arch=gfx950 -O3 tools/probe_gridbarrier.hip -o gpurun_out/probe_gridbarrier && ./probe_gridbarrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

// flat: one monotonically increasing counter; every WG adds 1 and spins until it reaches round * nwg
__global__ void __launch_bounds__(256) flat_kernel(unsigned* ctr, int rounds, unsigned* sink) {
    unsigned acc = 0;
    for (int r = 1; r <= rounds; ++r) {
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)r * gridDim.x;
            while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
        acc += r;
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) *sink = acc;
}
// hierarchical: WGs of one XCD (blockIdx % 8) meet on their own counter; the last arriver of each XCD bumps the root;
// everyone spins on a per-XCD release flag written by the WG that saw the root complete.
__global__ void __launch_bounds__(256) hier_kernel(unsigned* xcd_ctr /*[8*32]*/, unsigned* root, unsigned* flag /*[8*32]*/, int rounds, unsigned* sink) {
    const int x = blockIdx.x & 7;
    const unsigned per_xcd = gridDim.x / 8;
    unsigned acc = 0;
    for (int r = 1; r <= rounds; ++r) {
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned old = __hip_atomic_fetch_add(xcd_ctr + x * 32, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            if (old + 1 == (unsigned)r * per_xcd) {                 // last of this XCD
                const unsigned ro = __hip_atomic_fetch_add(root, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
                if (ro + 1 == (unsigned)r * 8) {                    // last XCD: release everybody
                    for (int k = 0; k < 8; ++k) __hip_atomic_store(flag + k * 32, (unsigned)r, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            while (__hip_atomic_load(flag + x * 32, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)r) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
        acc += r;
    }
    if (threadIdx.x == 0 && blockIdx.x == 0) *sink = acc;
}
__global__ void tiny_kernel(unsigned* sink) { if (threadIdx.x == 0 && blockIdx.x == 0) *sink += 1; }

int main() {
    unsigned* buf; hipMalloc(&buf, 1 << 16);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int rounds = 2000;
    for (int nwg : {256, 512, 1024}) {
        for (int variant = 0; variant < 2; ++variant) {
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                hipMemset(buf, 0, 1 << 16); hipDeviceSynchronize();
                hipEventRecord(a, 0);
                if (variant == 0) hipLaunchKernelGGL(flat_kernel, dim3(nwg), dim3(256), 0, 0, buf, rounds, buf + 8192);
                else hipLaunchKernelGGL(hier_kernel, dim3(nwg), dim3(256), 0, 0, buf, buf + 1024, buf + 2048, rounds, buf + 8192);
                hipEventRecord(b, 0); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                if (ms < best) best = ms;
            }
            printf("%s nwg=%4d : %.3f us per barrier\n", variant ? "hier" : "flat", nwg, best * 1e3f / rounds);
        }
    }
    // reference point: back-to-back tiny kernels in one stream
    hipEventRecord(a, 0);
    for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(tiny_kernel, dim3(256), dim3(256), 0, 0, buf + 8192);
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("tiny kernel launches (stream): %.3f us each\n", ms * 1e3f / 2000);
    // same through a graph
    hipStream_t st; hipStreamCreate(&st);
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < 500; ++i) hipLaunchKernelGGL(tiny_kernel, dim3(256), dim3(256), 0, st, buf + 8192);
    hipStreamEndCapture(st, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, st); hipStreamSynchronize(st);
    hipEventRecord(a, st); for (int i = 0; i < 4; ++i) hipGraphLaunch(ge, st); hipEventRecord(b, st); hipEventSynchronize(b);
    hipEventElapsedTime(&ms, a, b);
    printf("tiny kernel launches (graph): %.3f us each\n", ms * 1e3f / 2000);
    return 0;
}
