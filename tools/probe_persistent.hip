// Scratch micro-benchmark (not part of the product): is a PERSISTENT decode kernel worth building?
// One workgroup per CU stays resident and walks the step's launch groups ("phases": qkv 14.7 MB, attention 17.4 MB, wo 9.4 MB,
// gate/up 66 MB, down 40.6 MB per layer, 32 layers, every phase its own region of HBM) with a software grid barrier
// between phases.  Measures, per layer:
//   sep     : the same phases as separate launches in a hipGraph (what the product does today, without any compute)
//   nobar   : persistent, no barrier at all (upper bound of the memory path)
//   bar     : persistent, barrier between phases, the next phase's loads issued AFTER the barrier
//   pre     : persistent, barrier between phases, the next phase's first loads issued BEFORE the barrier (weights do not
//             depend on the previous phase's result)
//   pre+act : `pre` + a dependent hand-off: every wave publishes a word (write-through) before the barrier and reads 1 KiB
//             of other waves' words after it (sc1 loads) before it may consume its weights -- also CHECKS visibility
// and the cost of the barrier alone.  Barrier = relaxed agent-scope atomic add + sc1 polling (no release/acquire fences: an
// agent-scope release writes back the whole L2 and an acquire invalidates it -- that was the 11 us of probe_gridbarrier).
// Spins are bounded: a lost arrival sets an error flag and falls through (never hangs the box).
//   hipcc --offload-arch=gfx950 -O3 tools/probe_persistent.hip -o tools/probe_persistent.bin
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define WAVES 16
#define SPIN_LIMIT (1 << 22)

struct Phase { const u32x4* base; int lines; };

__device__ __forceinline__ void grid_barrier(unsigned* ctr, unsigned target, unsigned* err) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // this wave's write-through stores have left
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > SPIN_LIMIT) { atomicAdd(err, 1u); break; }
        }
    }
    __syncthreads();
}

__global__ void __launch_bounds__(64 * WAVES) barrier_only_kernel(unsigned* ctr, int rounds, unsigned* err, unsigned* sink) {
    extern __shared__ unsigned pad[];
    unsigned acc = 0;
    for (int r = 1; r <= rounds; ++r) { grid_barrier(ctr, (unsigned)r * gridDim.x, err); acc += r; }
    if (threadIdx.x == 0 && blockIdx.x == 0) *sink = acc + pad[0];
}

// MODE 0 nobar, 1 bar (issue after), 2 pre (issue before), 3 pre+act
template <int D, int MODE>
__global__ void __launch_bounds__(64 * WAVES) persistent_kernel(const Phase* __restrict__ ph, int nph, unsigned* ctr, unsigned* act,
                                                               unsigned* err, unsigned* sink) {
    extern __shared__ unsigned pad[];                                 // > 80 KB: one workgroup per CU
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wid = blockIdx.x * WAVES + wave, NWV = gridDim.x * WAVES;
    auto cnt = [&](int p) { const int n = ph[p].lines; const int c = n > wid ? (n - wid + NWV - 1) / NWV : 0; return (c + D - 1) / D * D; };
    auto addr = [&](int p, int i) {
        long line = (long)i * NWV + wid;
        if (line >= ph[p].lines) line = ph[p].lines - 1;              // padded slots re-read the last line (cache hit)
        return ph[p].base + line * 64 + lane;
    };
    u32x4 v[D];
    int ip = 0, il = 0;
    auto issue = [&](u32x4& dst, int limit_phase) {
        while (ip < nph && il >= cnt(ip)) { ++ip; il = 0; }
        if (ip < nph && ip <= limit_phase) { dst = __builtin_nontemporal_load(addr(ip, il)); ++il; }
    };
    unsigned acc = 0;
    constexpr bool AHEAD = MODE == 0 || MODE >= 2;
    if (AHEAD) {
#pragma unroll
        for (int u = 0; u < D; ++u) issue(v[u], nph);
    }
    for (int p = 0; p < nph; ++p) {
        if (MODE >= 1 && p > 0) {
            if (MODE == 3 && lane == 0) __hip_atomic_store(act + wid, (unsigned)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            grid_barrier(ctr, (unsigned)p * gridDim.x, err);
            if (MODE == 3) {
                const unsigned got = __hip_atomic_load(act + (wid * 61 + lane * 97) % NWV, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (got < (unsigned)p) atomicAdd(err + 1, 1u);       // a peer's pre-barrier store is not visible: protocol broken
                acc += got;
            }
        }
        const int n = cnt(p);
        if (!AHEAD) {
            ip = p; il = 0;
#pragma unroll
            for (int u = 0; u < D; ++u) issue(v[u], p);
        }
        for (int i = 0; i < n; i += D) {
#pragma unroll
            for (int u = 0; u < D; ++u) {
                acc ^= v[u].x ^ v[u].w;
                issue(v[u], AHEAD ? nph : p);
            }
        }
    }
    if (acc == 0x12345678u) sink[0] = acc + pad[lane];
}

template <int D>
__global__ void __launch_bounds__(256) stream_kernel(const u32x4* __restrict__ buf, unsigned* out, int n) {
    const int lane = threadIdx.x & 63;
    const long w = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const u32x4* p = buf + w * (long)n * 64 + lane;
    unsigned acc = 0;
    for (int i = 0; i < n; i += D) {
        u32x4 v[D];
#pragma unroll
        for (int u = 0; u < D; ++u) v[u] = __builtin_nontemporal_load(p + (long)(i + u < n ? i + u : n - 1) * 64);
#pragma unroll
        for (int u = 0; u < D; ++u) acc ^= v[u].x ^ v[u].w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at line %d\n", (int)e_, __LINE__); return 1; } } while (0)

int main() {
    int ncu = 0;
    CK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0));
    printf("CUs: %d\n", ncu);
    const int L = 32;
    const double mb[5] = {14.7, 17.4, 9.4, 66.0, 40.6};
    size_t off[5], per_layer = 0;
    for (int i = 0; i < 5; ++i) { off[i] = per_layer; per_layer += ((size_t)(mb[i] * 1e6) + 4095) / 4096 * 4096; }
    u32x4* buf; unsigned* small;
    CK(hipMalloc(&buf, per_layer * L));
    CK(hipMalloc(&small, 1 << 20));
    CK(hipMemset(buf, 1, per_layer * L));
    std::vector<Phase> ph;
    for (int l = 0; l < L; ++l)
        for (int i = 0; i < 5; ++i) ph.push_back(Phase{(const u32x4*)((const char*)buf + per_layer * l + off[i]), (int)(mb[i] * 1e6 / 1024)});
    Phase* dph;
    CK(hipMalloc(&dph, ph.size() * sizeof(Phase)));
    CK(hipMemcpy(dph, ph.data(), ph.size() * sizeof(Phase), hipMemcpyHostToDevice));
    unsigned *ctr = small, *err = small + 64, *sink = small + 128, *act = small + 1024;
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const size_t lds = 96 * 1024;
    CK(hipFuncSetAttribute((const void*)barrier_only_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    float ms;
    // ---- barrier alone
    for (int nwg : {ncu, ncu / 2}) {
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemsetAsync(small, 0, 1024, st));
            CK(hipEventRecord(a, st));
            hipLaunchKernelGGL(barrier_only_kernel, dim3(nwg), dim3(64 * WAVES), lds, st, ctr, 2000, err, sink);
            CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
            CK(hipEventElapsedTime(&ms, a, b));
            best = std::min(best, ms);
        }
        unsigned e[2]; CK(hipMemcpy(e, err, 8, hipMemcpyDeviceToHost));
        printf("barrier alone nwg=%d x %d thr: %.3f us per barrier (lost arrivals %u)\n", nwg, 64 * WAVES, best * 1e3f / 2000, e[0]);
    }
    // ---- persistent variants
#define RUN(D, MODE, name) do { \
        CK(hipFuncSetAttribute((const void*)persistent_kernel<D, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        std::vector<float> t; \
        for (int rep = 0; rep < 4; ++rep) { \
            CK(hipMemsetAsync(small, 0, 1 << 20, st)); \
            CK(hipEventRecord(a, st)); \
            hipLaunchKernelGGL((persistent_kernel<D, MODE>), dim3(ncu), dim3(64 * WAVES), lds, st, dph, (int)ph.size(), ctr, act, err, sink); \
            CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b)); \
            CK(hipEventElapsedTime(&ms, a, b)); t.push_back(ms); \
        } \
        std::sort(t.begin(), t.end()); \
        unsigned e[2]; CK(hipMemcpy(e, err, 8, hipMemcpyDeviceToHost)); \
        printf("persistent %-8s depth=%d : %7.2f us per layer  (%5.0f GB/s)  lost=%u stale=%u\n", name, D, t[1] * 1e3 / L, \
               per_layer * L / t[1] / 1e6, e[0], e[1]); \
    } while (0)
    RUN(2, 0, "nobar"); RUN(4, 0, "nobar");
    RUN(2, 1, "bar");   RUN(4, 1, "bar");
    RUN(2, 2, "pre");   RUN(4, 2, "pre");
    RUN(2, 3, "pre+act"); RUN(4, 3, "pre+act");
    // ---- the same phases as separate launches in a graph (18 lines per wave, depth 2: the best small-launch shape of probe_small)
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (auto& p : ph) {
        int lpw = p.lines >= 40000 ? 18 : (p.lines >= 12000 ? 4 : 2);
        const long wgs = p.lines / lpw / 4;
        hipLaunchKernelGGL(stream_kernel<2>, dim3(wgs), dim3(256), 0, st, p.base, sink, lpw);
    }
    CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
    std::vector<float> t;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(a, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(b, st)); CK(hipEventSynchronize(b));
        CK(hipEventElapsedTime(&ms, a, b)); t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    printf("separate launches in a graph     : %7.2f us per layer  (%5.0f GB/s)\n", t[1] * 1e3 / L, per_layer * L / t[1] / 1e6);
    return 0;
}
