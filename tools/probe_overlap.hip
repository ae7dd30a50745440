// Scratch micro-benchmark (not part of the product): can CONSECUTIVE launches of the single-token step overlap --
// launch N+1 dispatched without waiting for N, streaming its first weights while N drains, the dependency carried through device
// memory -- and what does that buy on a skeleton of the batch-1 step (same grids, same bytes, a dependent-FMA stand-in for the arithmetic)?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probe_overlap.hip -o build_probe/probe_overlap
//   build_probe/probe_overlap [fma_iters_per_load ...]          (AMD_LOG_LEVEL=4 build_probe/probe_overlap mini : the AQL headers)
// Hand-off forms:
//   CTR : the producer stores its outputs write-through (sc1), drains, adds 1 to an arrival counter (8 shards); one wave per consumer
//         workgroup polls the counters, then the workgroup loads the activations (sc1)
//   LL  : every output travels as an 8-byte {value, tag} pair stored write-through; the consumer polls THE DATA it needs until every
//         tag equals the producer's sequence number -- no drain, no atomic, no second round trip (the "LL" protocol of NCCL)
// Launch forms: ordinary (AQL barrier bit set; eager or hipGraph), any-order (hipExtLaunchKernel + hipExtAnyOrderLaunch: barrier bit
// clear -- eager only, stream capture drops the flag), two streams.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

enum { H_NONE = 0, H_CTR = 1, H_LL = 2 };

struct SkelArgs {
    const uint4* w;            // this launch's weight span
    int nkb;                   // k-blocks of the launch (x has nkb * 256 elements); wave v of a workgroup owns k-blocks v, v + nw, ...
    int lpk;                   // 1-KiB wave loads per k-block and wave (3 per Q4_K tile: 3 / 6)
    int fma;                   // dependent FMAs per load (stand-in for unpack + MFMA + scale)
    const void* x;             // activations: float[K] (CTR / ordinary) or {float, tag}[K] (LL)
    void* out;                 // what the launch produces: out_k elements in the consumer's format
    int out_k;
    int in_mode, out_mode;     // H_*
    unsigned in_tag, out_tag;  // LL tags
    unsigned* wait_ctr; unsigned wait_val; unsigned* sig_ctr;
    unsigned long long* ts;    // null or [6]: min start, max end, max wait-satisfied, max last-arrival, min wait-satisfied, -
    unsigned* err;
    int spin_sleep;
};

__device__ __forceinline__ unsigned long long now() { return wall_clock64(); }   // 100 MHz
typedef unsigned int u4_t __attribute__((ext_vector_type(4)));
typedef float f4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void spin_pause(int mode) {
    if (mode == 1) __builtin_amdgcn_s_sleep(8);
    else if (mode == 2) __builtin_amdgcn_s_sleep(32);
    else if (mode == 3) __builtin_amdgcn_s_sleep(127);
}

// the k-block's 256 activations, 4 per lane (ordinary / CTR formats)
__device__ __forceinline__ f4_t load_x(const SkelArgs& a, int kb, int lane) {
    f4_t v;
    if (a.in_mode == H_CTR) {
        const float* p = reinterpret_cast<const float*>(a.x) + (size_t)kb * 256 + lane * 4;
        asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    } else {
        v = *reinterpret_cast<const f4_t*>(reinterpret_cast<const float*>(a.x) + (size_t)kb * 256 + lane * 4);
    }
    return v;
}
// LL: ALL of the wave's k-blocks (<= 8) are polled together before the main loop -- one round trip per poll, and the loop keeps
// its counted waits (a poll inside the loop would drain the weight ring every k-block)
constexpr int LL_MAXKB = 8;
__device__ __forceinline__ void poll_x_ll(const SkelArgs& a, int wave, int nw, int n_my, int lane, f4_t (&xs)[LL_MAXKB]) {
    const unsigned long long tw = now();
    bool gave_up = __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
    for (;;) {
        u4_t p0[LL_MAXKB], p1[LL_MAXKB];
#pragma unroll
        for (int q = 0; q < LL_MAXKB; ++q) {
            const int kb = wave + nw * (q < n_my ? q : 0);
            const u4_t* p = reinterpret_cast<const u4_t*>(a.x) + ((size_t)kb * 256 + lane * 4) / 2;
            asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc1"
                         : "=&v"(p0[q]), "=&v"(p1[q]) : "v"(p) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        bool ok = true;
#pragma unroll
        for (int q = 0; q < LL_MAXKB; ++q) {
            asm volatile("" : "+v"(p0[q]), "+v"(p1[q]));
            ok = ok && p0[q].y == a.in_tag && p0[q].w == a.in_tag && p1[q].y == a.in_tag && p1[q].w == a.in_tag;
            xs[q] = f4_t{__uint_as_float(p0[q].x), __uint_as_float(p0[q].z), __uint_as_float(p1[q].x), __uint_as_float(p1[q].z)};
        }
        if (__all(ok) || gave_up) break;
        if (now() - tw > 2000000ull) { if (lane == 0) atomicAdd(a.err, 1u); break; }
        spin_pause(a.spin_sleep);
    }
}

template <int LPK>
__global__ void __launch_bounds__(512) skel(const SkelArgs a) {
    __shared__ float red[8][16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if (a.ts && threadIdx.x == 0) atomicMin(&a.ts[0], now());
    const int n_my = (a.nkb > wave) ? (a.nkb - wave + nw - 1) / nw : 0;
    // weights of (workgroup, k-block kb): LPK KiB, k-block-major inside the workgroup's span
    const u4_t* wp = reinterpret_cast<const u4_t*>(a.w) + ((size_t)blockIdx.x * a.nkb) * LPK * 64 + lane;
    constexpr int PFK = 2;                                          // ring depth in k-blocks (qmm_kernel: QMM_PF_MIN = 2)
    u4_t r[PFK][LPK];
#pragma unroll
    for (int q = 0; q < PFK; ++q) {
        const int kb = wave + nw * (q < n_my ? q : 0);
#pragma unroll
        for (int i = 0; i < LPK; ++i) r[q][i] = __builtin_nontemporal_load(wp + ((size_t)kb * LPK + i) * 64);
    }
    if (a.wait_ctr) {
        if (wave == 0) {
            const unsigned long long tw = now();
            const bool dead = __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
            for (; !dead;) {
                unsigned v = 0;
                if (lane < 8) v = __hip_atomic_load(&a.wait_ctr[lane * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
                v = __builtin_amdgcn_readfirstlane(v);
                if (v >= a.wait_val) break;
                if (now() - tw > 2000000ull) { if (lane == 0) atomicAdd(a.err, 1u); break; }   // 20 ms: give up, never hang
                spin_pause(a.spin_sleep);
            }
            if (a.ts && lane == 0) { const unsigned long long t = now(); atomicMax(&a.ts[2], t); atomicMin(&a.ts[4], t); }
        }
        __builtin_amdgcn_s_barrier();
    }
    f4_t xr[PFK];
    f4_t xs[LL_MAXKB];
    const bool ll = a.in_mode == H_LL;
    if (ll) {
        poll_x_ll(a, wave, nw, n_my, lane, xs);
        if (a.ts && threadIdx.x == 0) { const unsigned long long t = now(); atomicMax(&a.ts[2], t); atomicMin(&a.ts[4], t); }
    } else {
#pragma unroll
        for (int q = 0; q < PFK; ++q) xr[q] = load_x(a, wave + nw * (q < n_my ? q : 0), lane);
    }
    float acc = 0.f;
    for (int i0 = 0; i0 < n_my; i0 += PFK) {
#pragma unroll
        for (int q = 0; q < PFK; ++q) {
            const int i = i0 + q;
            if (i < n_my) {
                f4_t xv = xr[q];
                if (ll) {
#pragma unroll
                    for (int u = 0; u < LL_MAXKB; ++u) if (u == i) xv = xs[u];
                }
                float t = xv.x + xv.y + xv.z + xv.w;
#pragma unroll
                for (int j = 0; j < LPK; ++j) {
                    t += __uint_as_float(((r[q][j].x ^ r[q][j].y ^ r[q][j].z ^ r[q][j].w) & 0x007FFFFFu) | 0x3F800000u);
                    for (int f = 0; f < a.fma; ++f) t = fmaf(t, 0.999f, 0.001f);
                }
                acc += t;
            }
            const int n = i + PFK;
            const int kbn = wave + nw * (n < n_my ? n : 0);
            if (!ll) xr[q] = load_x(a, kbn, lane);
#pragma unroll
            for (int j = 0; j < LPK; ++j) r[q][j] = __builtin_nontemporal_load(wp + ((size_t)kbn * LPK + j) * 64);
        }
    }
    float y = acc + __shfl_xor(acc, 32, 64);
    if (lane < 16) red[wave][lane] = y;
    __syncthreads();
    // the workgroup's slice of what the consumer reads: elements [blockIdx * per, ...)
    const int per = (a.out_k + gridDim.x - 1) / gridDim.x;
    if ((int)threadIdx.x < per) {
        float s = 0.f;
        for (int w = 0; w < nw; ++w) s += red[w][threadIdx.x & 15];
        s = s * 1e-30f + 1.0f;
        const int e = blockIdx.x * per + threadIdx.x;
        if (e < a.out_k) {
            if (a.out_mode == H_LL) {
                const unsigned long long pv = (unsigned long long)__float_as_uint(s) | ((unsigned long long)a.out_tag << 32);
                __hip_atomic_store(reinterpret_cast<unsigned long long*>(a.out) + e, pv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else if (a.out_mode == H_CTR) {
                __hip_atomic_store(reinterpret_cast<float*>(a.out) + e, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                reinterpret_cast<float*>(a.out)[e] = s;
            }
        }
    }
    if (a.ts && a.out_mode != H_NONE && threadIdx.x == 0) atomicMax(&a.ts[3], now());
    if (a.sig_ctr) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                // (per <= 64 in every launch here: wave 0 stored everything)
        if (threadIdx.x == 0)
            __hip_atomic_fetch_add(&a.sig_ctr[(blockIdx.x & 7) * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (a.ts && threadIdx.x == 0) atomicMax(&a.ts[1], now());
}

__global__ void reset_kernel(unsigned* ctr, int n) { for (int i = threadIdx.x; i < n; i += blockDim.x) ctr[i] = 0; }
__global__ void ts_init_kernel(unsigned long long* ts, int n) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) ts[i] = ((i % 6) == 0 || (i % 6) == 4) ? ~0ull : 0ull;
}

struct LaunchSpec { const char* name; int n_wg, nw, nkb, lpk; bool attn; };
// the batch-1 step of Llama-3-8B Q4_K_M as qmm_kernel launches it (grids from mi355_qmm_launch / qmm_pick_nw); attention as a 17 MB stream
static const LaunchSpec kLayer[5] = {
    {"qkv", 384, 8, 16, 3, false}, {"attn", 136, 8, 16, 8, true}, {"wo", 256, 8, 16, 3, false},
    {"gateup", 896, 2, 16, 6, false}, {"down", 256, 8, 56, 3, false}};

enum Mode { M_NORMAL = 0, M_CTR_ALL = 1, M_LL_ALL = 2, M_LL = 3, M_CTR = 4, M_LL_TWOSTREAM = 5, M_LL_ORDERED = 6 };
static const char* kModeName[] = {"ordinary", "any-order + counters (all 5)", "any-order + LL (all 5)", "any-order + LL (attention ordinary)",
                                  "any-order + counters (attention ordinary)", "two streams + LL", "ordinary launches + LL format (control)"};

struct Ctx {
    uint4* w; size_t w_bytes; void* x[2]; unsigned* ctr; unsigned long long* ts; unsigned* err;
    hipStream_t s[2];
    int spin_sleep, fma;
    unsigned seq;
};

static hipError_t enqueue(Ctx& c, Mode mode, int layers, bool with_ts) {
    size_t woff = 0;
    int k = 0;
    const bool ctr = mode == M_CTR_ALL || mode == M_CTR;
    const bool ll = mode == M_LL_ALL || mode == M_LL || mode == M_LL_TWOSTREAM || mode == M_LL_ORDERED;
    if (ctr) hipLaunchKernelGGL(reset_kernel, dim3(1), dim3(256), 0, c.s[0], c.ctr, layers * 5 * 8 * 32);
    if (mode == M_LL_TWOSTREAM) {
        hipEvent_t e; CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        CHECK(hipEventRecord(e, c.s[0])); CHECK(hipStreamWaitEvent(c.s[1], e, 0)); CHECK(hipEventDestroy(e));
    }
    for (int l = 0; l < layers; ++l)
        for (int j = 0; j < 5; ++j, ++k) {
            const LaunchSpec& L = kLayer[j];
            const LaunchSpec& N = kLayer[(j + 1) % 5];
            const LaunchSpec& P = kLayer[(j + 4) % 5];
            const size_t bytes = (size_t)L.n_wg * L.nkb * L.lpk * 1024;
            if (woff + bytes > c.w_bytes) woff = 0;
            SkelArgs a{};
            a.w = c.w + woff / 16; woff += bytes;
            a.nkb = L.nkb; a.lpk = L.lpk; a.fma = c.fma;
            a.x = c.x[k & 1]; a.out = c.x[(k + 1) & 1]; a.out_k = N.nkb * 256;
            a.err = c.err; a.spin_sleep = c.spin_sleep;
            a.ts = with_ts ? c.ts + (size_t)k * 6 : nullptr;
            const bool first = (k == 0);
            const bool ordinary = mode == M_NORMAL || mode == M_LL_ORDERED || first || ((mode == M_LL || mode == M_CTR) && L.attn);
            a.in_mode = H_NONE; a.out_mode = H_NONE;
            if (ctr) {
                a.out_mode = H_CTR; a.sig_ctr = c.ctr + (size_t)k * 8 * 32;
                if (!ordinary) { a.in_mode = H_CTR; a.wait_ctr = c.ctr + (size_t)(k - 1) * 8 * 32; a.wait_val = (unsigned)P.n_wg; }
                else if (!first) a.in_mode = H_CTR;                  // an ordinary launch behind sc1 stores: plain loads would do, keep the format
            }
            if (ll) {
                a.out_mode = H_LL; a.out_tag = ++c.seq;
                if (!first) { a.in_mode = H_LL; a.in_tag = c.seq - 1; }
            }
            hipStream_t st = (mode == M_LL_TWOSTREAM) ? c.s[k & 1] : c.s[0];
#define GO(LPK_) do { \
                if (!ordinary && mode != M_LL_TWOSTREAM) hipExtLaunchKernelGGL((skel<LPK_>), dim3(L.n_wg), dim3(64 * L.nw), 0, st, nullptr, nullptr, hipExtAnyOrderLaunch, a); \
                else hipLaunchKernelGGL((skel<LPK_>), dim3(L.n_wg), dim3(64 * L.nw), 0, st, a); } while (0)
            if (L.lpk == 3) GO(3); else if (L.lpk == 6) GO(6); else GO(8);
#undef GO
        }
    if (mode == M_LL_TWOSTREAM) {
        hipEvent_t e; CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        CHECK(hipEventRecord(e, c.s[1])); CHECK(hipStreamWaitEvent(c.s[0], e, 0)); CHECK(hipEventDestroy(e));
    }
    return hipGetLastError();
}

static unsigned read_err(Ctx& c) { unsigned e; CHECK(hipMemcpy(&e, c.err, 4, hipMemcpyDeviceToHost)); return e; }

static void timeline(Ctx& c, Mode mode, int layers, bool graph) {
    const char* tag = kModeName[mode];
    CHECK(hipMemset(c.err, 0, 4));
    hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(ts_init_kernel, dim3(1), dim3(256), 0, c.s[0], c.ts, layers * 5 * 6);
        CHECK(hipStreamSynchronize(c.s[0]));
        if (graph) {
            if (g) { CHECK(hipGraphExecDestroy(ge)); CHECK(hipGraphDestroy(g)); }
            CHECK(hipStreamBeginCapture(c.s[0], hipStreamCaptureModeThreadLocal));
            const hipError_t ce = enqueue(c, mode, layers, true);
            const hipError_t ee = hipStreamEndCapture(c.s[0], &g);
            if (ce != hipSuccess || ee != hipSuccess) { printf("timeline %s: CAPTURE FAILED\n", tag); (void)hipGetLastError(); return; }
            CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            CHECK(hipGraphLaunch(ge, c.s[0]));
        } else {
            CHECK(enqueue(c, mode, layers, true));
        }
        CHECK(hipStreamSynchronize(c.s[0])); CHECK(hipStreamSynchronize(c.s[1]));
    }
    if (g) { CHECK(hipGraphExecDestroy(ge)); CHECK(hipGraphDestroy(g)); }
    std::vector<unsigned long long> ts((size_t)layers * 5 * 6);
    CHECK(hipMemcpy(ts.data(), c.ts, ts.size() * 8, hipMemcpyDeviceToHost));
    const unsigned long long t0 = ts[0];
    printf("timeline [%s] %s (us from the first kernel's start; per-workgroup timestamps cost time: read the order, not the totals):\n",
           graph ? "graph" : "eager", tag);
    for (int k = 0; k < layers * 5 && k < 10; ++k) {
        const unsigned long long* t = &ts[(size_t)k * 6];
        printf("  k=%2d %-6s start %7.2f  x-complete first %7.2f last %7.2f  last-store %7.2f  end %7.2f%s\n", k, kLayer[k % 5].name,
               (t[0] - t0) * 0.01, t[2] ? (t[4] - t0) * 0.01 : 0.0, t[2] ? (t[2] - t0) * 0.01 : 0.0, t[3] ? (t[3] - t0) * 0.01 : 0.0,
               (t[1] - t0) * 0.01, (k > 0 && t[0] < ts[(size_t)(k - 1) * 6 + 1]) ? "   STARTS INSIDE its predecessor" : "");
    }
    printf("  err=%u\n", read_err(c));
}

static double time_mode(Ctx& c, Mode mode, int layers, bool graph, int reps) {
    CHECK(hipMemset(c.err, 0, 4));                                  // a give-up of an earlier mode must not switch this one's waits off
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float ms = 0.f;
    if (graph) {
        // LL tags are kernel arguments: a graph replays the same tags -- only the tag-free modes can be replayed
        hipGraph_t g; hipGraphExec_t ge;
        CHECK(hipStreamBeginCapture(c.s[0], hipStreamCaptureModeThreadLocal));
        const hipError_t ce = enqueue(c, mode, layers, false);
        const hipError_t ee = hipStreamEndCapture(c.s[0], &g);
        if (ce != hipSuccess || ee != hipSuccess) { (void)hipGetLastError(); return -1.0; }
        CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int i = 0; i < 3; ++i) CHECK(hipGraphLaunch(ge, c.s[0]));
        CHECK(hipStreamSynchronize(c.s[0]));
        CHECK(hipEventRecord(e0, c.s[0]));
        for (int i = 0; i < reps; ++i) CHECK(hipGraphLaunch(ge, c.s[0]));
        CHECK(hipEventRecord(e1, c.s[0])); CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        CHECK(hipGraphExecDestroy(ge)); CHECK(hipGraphDestroy(g));
    } else {
        for (int i = 0; i < 3; ++i) CHECK(enqueue(c, mode, layers, false));
        CHECK(hipStreamSynchronize(c.s[0])); CHECK(hipStreamSynchronize(c.s[1]));
        CHECK(hipEventRecord(e0, c.s[0]));
        for (int i = 0; i < reps; ++i) CHECK(enqueue(c, mode, layers, false));
        CHECK(hipEventRecord(e1, c.s[0])); CHECK(hipEventSynchronize(e1));
        CHECK(hipStreamSynchronize(c.s[1]));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
    }
    CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
    return ms * 1e3 / reps / layers;
}

int main(int argc, char** argv) {
    const bool mini = argc > 1 && !strcmp(argv[1], "mini");      // one layer (run under AMD_LOG_LEVEL=4 to read the AQL headers)
    Ctx c{};
    c.w_bytes = 3ull << 30;
    CHECK(hipMalloc(&c.w, c.w_bytes)); CHECK(hipMemset(c.w, 1, c.w_bytes));
    CHECK(hipMalloc(&c.x[0], 1 << 20)); CHECK(hipMalloc(&c.x[1], 1 << 20));
    CHECK(hipMemset(c.x[0], 0, 1 << 20)); CHECK(hipMemset(c.x[1], 0, 1 << 20));
    CHECK(hipMalloc(&c.ctr, 32 * 5 * 8 * 32 * 4 + 4096)); CHECK(hipMalloc(&c.ts, 32 * 5 * 6 * 8)); CHECK(hipMalloc(&c.err, 4));
    CHECK(hipMemset(c.err, 0, 4));
    CHECK(hipStreamCreateWithFlags(&c.s[0], hipStreamNonBlocking)); CHECK(hipStreamCreateWithFlags(&c.s[1], hipStreamNonBlocking));
    c.spin_sleep = 1; c.fma = 40; c.seq = 100;
    if (mini) {
        CHECK(enqueue(c, M_LL_ALL, 1, false));
        CHECK(hipStreamSynchronize(c.s[0]));
        printf("mini done err=%u\n", read_err(c));
        return 0;
    }
    std::vector<int> fmas;
    for (int i = 1; i < argc; ++i) fmas.push_back(atoi(argv[i]));
    if (fmas.empty()) fmas = {0, 40};
    for (int fma : fmas) {
        c.fma = fma;
        printf("==== %d dependent FMAs per KiB load\n", fma);
        timeline(c, M_NORMAL, 2, false);
        timeline(c, M_NORMAL, 2, true);
        timeline(c, M_CTR_ALL, 2, false);
        timeline(c, M_LL_ALL, 2, false);
        timeline(c, M_LL, 2, false);
        for (int sl : {1, 2, 0}) {
            c.spin_sleep = sl;
            printf("chain of 32 layers, us per layer (spin pause %d):", sl);
            printf("  ordinary eager %.2f", time_mode(c, M_NORMAL, 32, false, 10));
            printf("  ordinary graph %.2f", time_mode(c, M_NORMAL, 32, true, 10));
            printf("  | LL format, ordinary eager %.2f", time_mode(c, M_LL_ORDERED, 32, false, 10));
            printf("  | any-order: counters all %.2f", time_mode(c, M_CTR_ALL, 32, false, 10));
            printf("  counters attn-ordinary %.2f", time_mode(c, M_CTR, 32, false, 10));
            printf("  LL all %.2f", time_mode(c, M_LL_ALL, 32, false, 10));
            printf("  LL attn-ordinary %.2f", time_mode(c, M_LL, 32, false, 10));
            printf("  (give-ups %u)", read_err(c));
            printf("  | two streams LL %.2f (give-ups %u)\n", time_mode(c, M_LL_TWOSTREAM, 32, false, 10), read_err(c));
            fflush(stdout);
        }
        c.spin_sleep = 1;
    }
    return 0;
}
