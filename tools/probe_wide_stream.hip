// Scratch micro-benchmark (not part of the product): what bounds the bare weight stream of the wide (9..32-token) GEMM?
// Replays the load skeleton of qmm_gemm_kernel for the gate/up launch of Llama-3-8B (1792 row tiles x 16 k-blocks x 2304 B
// = 66 MB, one 36 KB stream per tile) with one thing changed at a time.  Six distinct weight buffers are cycled so the
// Infinity Cache never holds the one being read.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_wide_stream.hip -o tools/probe_wide_stream.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define TILE_B 2304
#define NKB 16
#define NT 1792

template <bool NTL>
__device__ __forceinline__ u32x4 ld16(const uint8_t* p) {
    if (NTL) return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
    return *reinterpret_cast<const u32x4*>(p);
}
struct Tile { u32x4 a, b, c; };
template <bool NTL, bool DUP>
__device__ __forceinline__ Tile ld_tile(const uint8_t* t, int lane) {
    Tile r;
    r.a = ld16<NTL>(t + (DUP ? (lane & 15) : (lane & 15)) * 16);
    r.b = ld16<NTL>(t + 256 + lane * 16);
    r.c = ld16<NTL>(t + 1280 + lane * 16);
    return r;
}
__device__ __forceinline__ unsigned fold(const Tile& t) { return t.a.x ^ t.b.y ^ t.c.z; }

// WAVES waves per workgroup, LOADERS of them stream one tile each; KS = k-splits (gridDim.y); PF tiles in flight per wave;
// BAR = a workgroup barrier per k-block (the product has one: the shared activation image is double-buffered in LDS)
template <int WAVES, int LOADERS, int PF, bool BAR, bool NTL, int LDSKB>
__global__ void __launch_bounds__(64 * WAVES) wide_like(const uint8_t* __restrict__ w, unsigned* __restrict__ sink, int n_tiles, int tile_stride, int kb_stride) {
    extern __shared__ uint8_t smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kb_per = NKB / gridDim.y, kb_lo = blockIdx.y * kb_per, kb_hi = kb_lo + kb_per;
    int slot = blockIdx.x * LOADERS + wave;
    const bool have = wave < LOADERS && slot < n_tiles;
    if (!have) slot = 0;
    const uint8_t* base = w + (size_t)slot * tile_stride;
    unsigned acc = 0;
    if (LDSKB && threadIdx.x == 0) smem[0] = 1;
    if (wave >= LOADERS) {
        if (BAR) for (int kb = kb_lo; kb <= kb_hi; ++kb) __syncthreads();
        return;
    }
    Tile q[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i) q[i] = ld_tile<NTL, true>(base + (size_t)min(kb_lo + i, kb_hi - 1) * kb_stride, lane);
    if (BAR) __syncthreads();
    for (int kb = kb_lo; kb < kb_hi; ++kb) {
        Tile far = ld_tile<NTL, true>(base + (size_t)min(kb + PF, kb_hi - 1) * kb_stride, lane);
        acc += fold(q[0]);
#pragma unroll
        for (int i = 0; i + 1 < PF; ++i) q[i] = q[i + 1];
        q[PF - 1] = far;
        if (BAR) __syncthreads();
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

// the geometry of the 1..8-token kernel: one workgroup per tile, NW waves take every NW-th k-block, two in flight
template <int NW>
__global__ void __launch_bounds__(64 * NW) narrow_like(const uint8_t* __restrict__ w, unsigned* __restrict__ sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint8_t* base = w + (size_t)blockIdx.x * NKB * TILE_B;
    unsigned acc = 0;
    Tile cur = ld_tile<true, true>(base + (size_t)wave * TILE_B, lane);
    for (int kb = wave; kb < NKB; kb += NW) {
        const int kn = kb + NW < NKB ? kb + NW : kb;
        Tile nxt = ld_tile<true, true>(base + (size_t)kn * TILE_B, lane);
        acc += fold(cur);
        cur = nxt;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}


typedef const void __attribute__((address_space(1)))* gptr_t;
typedef void __attribute__((address_space(3)))* lptr_t;
// closer to the product: wave 7 is the loader of the shared activation image (38 KB per k-block, LDS DMA, double buffer),
// the consumers store f32 partial sums [k-split][32 tokens][rows] the way the product does
template <bool DMA, bool STORE, bool LDSREAD>
__global__ void __launch_bounds__(512, 4) product_like(const uint8_t* __restrict__ w, const uint8_t* __restrict__ img, float* __restrict__ part,
                                                       int n_tiles, int ldp) {
    extern __shared__ __attribute__((aligned(1024))) uint8_t smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr size_t kbb = 38912;
    const int kb_per = NKB / gridDim.y, kb_lo = blockIdx.y * kb_per, kb_hi = kb_lo + kb_per;
    if (wave == 7) {
        auto dma = [&](int kb, int buf) {
            if (!DMA) return;
            const uint8_t* src = img + (size_t)kb * kbb;
            uint8_t* dst = smem + (size_t)buf * kbb;
            for (int c = 0; c < 38; ++c)
                __builtin_amdgcn_global_load_lds((gptr_t)(src + (size_t)c * 1024 + lane * 16), (lptr_t)(dst + (size_t)c * 1024), 16, 0, 0);
        };
        dma(kb_lo, kb_lo & 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int kb = kb_lo; kb < kb_hi; ++kb) {
            if (kb + 1 < kb_hi) dma(kb + 1, (kb + 1) & 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        return;
    }
    int slot = blockIdx.x * 7 + wave;
    const bool have = slot < n_tiles;
    if (!have) slot = 0;
    const uint8_t* base = w + (size_t)slot * NKB * TILE_B;
    float y[4][4];
    for (int mt = 0; mt < 4; ++mt) for (int v = 0; v < 4; ++v) y[mt][v] = 0.f;
    Tile cur = ld_tile<true, true>(base + (size_t)kb_lo * TILE_B, lane);
    __syncthreads();
    for (int kb = kb_lo; kb < kb_hi; ++kb) {
        const bool more = kb + 1 < kb_hi;
        Tile nxt = ld_tile<true, true>(more ? base + (size_t)(kb + 1) * TILE_B : base, more ? lane : 0);
        y[0][0] += __uint_as_float(fold(cur) & 0x3fffffffu);
        if (LDSREAD) {                                                 // the product reads 32 KB of the image per wave and k-block
            const uint8_t* L = smem + (size_t)(kb & 1) * kbb;
            for (int j = 0; j < 32; ++j) {
                const uint4 a4 = *reinterpret_cast<const uint4*>(L + (size_t)j * 1024 + lane * 16);
                y[j & 3][(j >> 2) & 3] += __uint_as_float(a4.x & 0x3fffffffu);
            }
        }
        cur = nxt;
        __syncthreads();
    }
    const int kg = lane >> 4, rr = lane & 15;
    if (STORE && have && kg < 2) {
        float* pp = part + (size_t)blockIdx.y * 32 * ldp + (size_t)slot * 16 + rr;
        for (int mt = 0; mt < 4; ++mt)
            for (int v = 0; v < 4; ++v) pp[(size_t)(8 * mt + 4 * kg + v) * ldp] = y[mt][v];
    } else if (y[0][0] == 1.2345f) part[0] = y[1][1] + y[2][2] + y[3][3];
}


// round 4: the same skeleton with the image of the fourth generation (IKB = 18 KiB per k-block: one f16 plane) and a ring of NBUF stages
// filled NBUF - 1 k-blocks ahead (counted vmcnt + a bare s_barrier in the loader: the fence of __syncthreads would drain the DMAs)
template <int IKB, int NBUF, bool STORE, bool LDSREAD>
__global__ void __launch_bounds__(512, 4) product_like2(const uint8_t* __restrict__ w, const uint8_t* __restrict__ img, float* __restrict__ part,
                                                        int n_tiles, int ldp) {
    extern __shared__ __attribute__((aligned(1024))) uint8_t smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr size_t kbb = (size_t)IKB * 1024;
    constexpr int LEAD = NBUF - 1;
    const int kb_per = NKB / gridDim.y, kb_lo = blockIdx.y * kb_per, kb_hi = kb_lo + kb_per;
    if (wave == 7) {
        auto dma = [&](int kb, int buf) {
            const uint8_t* src = img + (size_t)kb * kbb;
            uint8_t* dst = smem + (size_t)buf * kbb;
#pragma unroll
            for (int c = 0; c < IKB; ++c)
                __builtin_amdgcn_global_load_lds((gptr_t)(src + (size_t)c * 1024 + lane * 16), (lptr_t)(dst + (size_t)c * 1024), 16, 0, 0);
        };
        for (int i = 0; i < LEAD; ++i) if (kb_lo + i < kb_hi) dma(kb_lo + i, i);
        if (LEAD == 2 && kb_lo + 1 < kb_hi) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IKB) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        int bufn = LEAD % NBUF;
        for (int kb = kb_lo; kb < kb_hi; ++kb) {
            const bool req = kb + LEAD < kb_hi;
            if (req) dma(kb + LEAD, bufn);
            bufn = bufn + 1 == NBUF ? 0 : bufn + 1;
            if (LEAD == 2 && req) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IKB) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        return;
    }
    int slot = blockIdx.x * 7 + wave;
    const bool have = slot < n_tiles;
    if (!have) slot = 0;
    const uint8_t* base = w + (size_t)slot * NKB * TILE_B;
    float y[4][4];
    for (int mt = 0; mt < 4; ++mt) for (int v = 0; v < 4; ++v) y[mt][v] = 0.f;
    Tile cur = ld_tile<true, true>(base + (size_t)kb_lo * TILE_B, lane);
    __syncthreads();
    int buf = 0;
    for (int kb = kb_lo; kb < kb_hi; ++kb) {
        const bool more = kb + 1 < kb_hi;
        Tile nxt = ld_tile<true, true>(more ? base + (size_t)(kb + 1) * TILE_B : base, more ? lane : 0);
        y[0][0] += __uint_as_float(fold(cur) & 0x3fffffffu);
        if (LDSREAD) {                                                 // the product reads 16 KB of the image per wave and k-block
            const uint8_t* L = smem + (size_t)buf * kbb;
            for (int j = 0; j < 16; ++j) {
                const uint4 a4 = *reinterpret_cast<const uint4*>(L + (size_t)j * 1024 + lane * 16);
                y[j & 3][(j >> 2) & 3] += __uint_as_float(a4.x & 0x3fffffffu);
            }
        }
        buf = buf + 1 == NBUF ? 0 : buf + 1;
        cur = nxt;
        __syncthreads();
    }
    const int kg = lane >> 4, rr = lane & 15;
    if (STORE && have && kg < 2) {
        float* pp = part + (size_t)blockIdx.y * 32 * ldp + (size_t)slot * 16 + rr;
        for (int mt = 0; mt < 4; ++mt)
            for (int v = 0; v < 4; ++v) pp[(size_t)(8 * mt + 4 * kg + v) * ldp] = y[mt][v];
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main() {
    const size_t bytes = (size_t)NT * NKB * TILE_B;
    const int NBUF = 6, REPS = 30;
    std::vector<uint8_t*> bufs(NBUF);
    for (auto& b : bufs) { CK(hipMalloc(&b, bytes)); CK(hipMemset(b, 0x5a, bytes)); }
    unsigned* sink; CK(hipMalloc(&sink, 64));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, auto launch) -> int {
        for (int i = 0; i < 6; ++i) launch(bufs[i % NBUF]);
        CK(hipStreamSynchronize(st));
        float best = 1e9f, sum = 0.f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < REPS; ++i) launch(bufs[i % NBUF]);
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best; sum += ms;
        }
        const double us = best * 1e3 / REPS;
        printf("%-64s %8.2f us/launch  %6.2f TB/s\n", name, us, bytes / us * 1e-6);
        fflush(stdout);
        return 0;
    };
    const int TS = NKB * TILE_B, KS_ = TILE_B;                           // product layout: [tile][k-block][2304 B]
#define WIDE(name, WAVES, LOADERS, PF, BAR, NTL, LDSKB, KSPLIT) \
    { CK(hipFuncSetAttribute((const void*)wide_like<WAVES, LOADERS, PF, BAR, NTL, LDSKB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
      if (timeit(name, [&](uint8_t* b) { hipLaunchKernelGGL((wide_like<WAVES, LOADERS, PF, BAR, NTL, LDSKB>), dim3((NT + LOADERS - 1) / LOADERS, KSPLIT), dim3(64 * WAVES), LDSKB * 1024, st, b, sink, NT, TS, KS_); })) return 1; }
    WIDE("product skeleton: 8 waves/7 load, barrier, PF1, 76 KB LDS", 8, 7, 1, true, true, 76, 1)
    WIDE("  no barrier", 8, 7, 1, false, true, 76, 1)
    WIDE("  no LDS (several workgroups per CU allowed)", 8, 7, 1, true, true, 0, 1)
    WIDE("  no LDS, no barrier", 8, 7, 1, false, true, 0, 1)
    WIDE("  PF2, barrier, 76 KB LDS", 8, 7, 2, true, true, 76, 1)
    WIDE("  PF4, barrier, 76 KB LDS", 8, 7, 4, true, true, 76, 1)
    WIDE("  PF4, no barrier, 76 KB LDS", 8, 7, 4, false, true, 76, 1)
    WIDE("  PF4, no barrier, no LDS", 8, 7, 4, false, true, 0, 1)
    WIDE("  temporal loads, barrier, PF1, 76 KB", 8, 7, 1, true, false, 76, 1)
    WIDE("  k split 2 (512 workgroups), barrier, PF1, 76 KB", 8, 7, 1, true, true, 76, 2)
    WIDE("  k split 4 (1024 workgroups), barrier, PF1, 76 KB", 8, 7, 1, true, true, 76, 4)
    WIDE("  16 waves/14 load (128 workgroups), barrier, PF1, 76 KB", 16, 14, 1, true, true, 76, 1)
    WIDE("  16 waves/14 load, PF4, 76 KB", 16, 14, 4, true, true, 76, 1)
    WIDE("  4 waves/4 load (448 workgroups), barrier, PF1, 38 KB", 4, 4, 1, true, true, 38, 1)
    WIDE("  4 waves/4 load, PF4, 38 KB", 4, 4, 4, true, true, 38, 1)
    WIDE("  2 waves/2 load (896 workgroups), PF4, 19 KB", 2, 2, 4, true, true, 19, 1)
    if (timeit("narrow geometry: 1792 workgroups x 4 waves, waves split k", [&](uint8_t* b) { hipLaunchKernelGGL((narrow_like<4>), dim3(NT), dim3(256), 0, st, b, sink); })) return 1;
    if (timeit("narrow geometry: 1792 workgroups x 8 waves", [&](uint8_t* b) { hipLaunchKernelGGL((narrow_like<8>), dim3(NT), dim3(512), 0, st, b, sink); })) return 1;

    uint8_t* img; CK(hipMalloc(&img, (size_t)NKB * 38912)); CK(hipMemset(img, 0x11, (size_t)NKB * 38912));
    float* part; CK(hipMalloc(&part, (size_t)4 * 32 * NT * 16 * sizeof(float)));
#define PROD(name, DMA, STORE, LDSREAD, KSPLIT) \
    { CK(hipFuncSetAttribute((const void*)product_like<DMA, STORE, LDSREAD>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
      if (timeit(name, [&](uint8_t* b) { hipLaunchKernelGGL((product_like<DMA, STORE, LDSREAD>), dim3(NT / 7, KSPLIT), dim3(512), 2 * 38912, st, b, img, part, NT, NT * 16); })) return 1; }
    PROD("product-like: loader idle, no stores", false, false, false, 1)
    PROD("  + partial-sum stores", false, true, false, 1)
    PROD("  + image DMA (38 KB per k-block and workgroup)", true, false, false, 1)
    PROD("  + both", true, true, false, 1)
    PROD("  + both + 32 KB of LDS reads per wave and k-block", true, true, true, 1)
    PROD("  all, k split 2", true, true, true, 2)
#define PROD2(name, IKB, NBUF, STORE, LDSREAD, KSPLIT) \
    { CK(hipFuncSetAttribute((const void*)product_like2<IKB, NBUF, STORE, LDSREAD>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
      if (timeit(name, [&](uint8_t* b) { hipLaunchKernelGGL((product_like2<IKB, NBUF, STORE, LDSREAD>), dim3(NT / 7, KSPLIT), dim3(512), (size_t)NBUF * IKB * 1024, st, b, img, part, NT, NT * 16); })) return 1; }
    PROD2("round 4 image (18 KB per k-block): DMA, double buffer", 18, 2, false, false, 1)
    PROD2("  + stores + 16 KB of LDS reads per wave and k-block", 18, 2, true, true, 1)
    PROD2("  three stages, two k-blocks ahead: DMA only", 18, 3, false, false, 1)
    PROD2("  three stages + stores + LDS reads", 18, 3, true, true, 1)
    PROD2("  38 KB image, three stages, DMA only", 38, 3, false, false, 1)
    PROD2("  18 KB, double buffer, all, k split 2", 18, 2, true, true, 2)
    PROD2("  18 KB, three stages, all, k split 2", 18, 3, true, true, 2)
    return 0;
}
