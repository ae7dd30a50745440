// Tensor-parallel collectives of the host layers.
//   reference: one nccl `Comm` per process (src/openai/pipelines/pipeline.rs:805-812); AllReduce / AllGather custom ops
//   (src/openai/distributed.rs:547-654,1335-1446) called after o_proj / down_proj (attention.rs:1003-1008,
//   quantized_llama.rs:38-42) and for the vocab-parallel lm_head (distributed.rs:1632-1667).
// Three transports behind one handle:
//   * RCCL (bound by dlopen, so the library still loads on a CPU-only box): enqueued on a SIDE stream fenced by two
//     events, the step stream never carries a collective itself (north_star);
//   * a one-shot peer-to-peer all-reduce for decode-sized messages (<= 256 KiB): every rank publishes its partial in a
//     fine-grained region its peers have opened through an IPC handle, then sums all partials itself in rank order --
//     one kernel, no ring hops (xGMI is point-to-point: a ring all-reduce of 16 KiB is 2(W-1) link latencies), bit-identical
//     results on every rank, and replayable inside a hipGraph (sequence numbers live on the device);
//   * collectives supplied by the host (it already owns a communicator, as the reference's Rust side does).
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

#include "comm.h"
#include "common.h"
#include "scratch.h"

namespace {
struct NcclId { char internal[128]; };
typedef int (*nccl_get_id_t)(NcclId*);
typedef int (*nccl_init_rank_t)(void**, int, NcclId, int);
typedef int (*nccl_allreduce_t)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*nccl_allgather_t)(const void*, void*, size_t, int, void*, hipStream_t);
typedef int (*nccl_destroy_t)(void*);
struct Rccl {
    void* h = nullptr;
    nccl_get_id_t get_id = nullptr;
    nccl_init_rank_t init_rank = nullptr;
    nccl_allreduce_t all_reduce = nullptr;
    nccl_allgather_t all_gather = nullptr;
    nccl_destroy_t destroy = nullptr;
};
Rccl g_rccl;
bool rccl_load() {
    if (g_rccl.h) return true;
    const char* env = getenv("MI355_RCCL_PATH");
    const char* names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void* h = nullptr;
    for (int pass = 0; pass < 2 && !h; ++pass)            // pass 0: only a copy that is already loaded (torch's)
        for (const char* n : names) {
            if (!n) continue;
            h = dlopen(n, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
            if (h) break;
        }
    if (!h) return false;
    g_rccl.get_id = (nccl_get_id_t)dlsym(h, "ncclGetUniqueId");
    g_rccl.init_rank = (nccl_init_rank_t)dlsym(h, "ncclCommInitRank");
    g_rccl.all_reduce = (nccl_allreduce_t)dlsym(h, "ncclAllReduce");
    g_rccl.all_gather = (nccl_allgather_t)dlsym(h, "ncclAllGather");
    g_rccl.destroy = (nccl_destroy_t)dlsym(h, "ncclCommDestroy");
    if (!g_rccl.get_id || !g_rccl.init_rank || !g_rccl.all_reduce || !g_rccl.all_gather) return false;
    g_rccl.h = h;
    return true;
}
enum { NCCL_SUM = 0 };
int nccl_dtype_of(int dt) { return dt == MI355_DTYPE_F32 ? 7 : (dt == MI355_DTYPE_F16 ? 6 : (dt == MI355_DTYPE_BF16 ? 9 : -1)); }

#define CCHECK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return (int)e_; } while (0)

// RCCL call on the side stream: step stream -> event -> side stream [collective] -> event -> step stream.  Inside a stream
// capture this is the fork / join pattern, so the collective becomes a node of the step's graph.
template <class F>
int on_side_stream(Comm* c, hipStream_t st, F enqueue) {
    if (!c->side) return enqueue(st);
    if (!c->cs) {
        CCHECK(hipStreamCreateWithFlags(&c->cs, hipStreamNonBlocking));
        CCHECK(hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming));
        CCHECK(hipEventCreateWithFlags(&c->ev_out, hipEventDisableTiming));
    }
    CCHECK(hipEventRecord(c->ev_in, st));
    CCHECK(hipStreamWaitEvent(c->cs, c->ev_in, 0));
    const int rc = enqueue(c->cs);
    if (rc) return rc;
    CCHECK(hipEventRecord(c->ev_out, c->cs));
    CCHECK(hipStreamWaitEvent(st, c->ev_out, 0));
    return 0;
}

__device__ __forceinline__ float bf16_round_f(float f) { return bf16_to_f32(f32_to_bf16(f)); }

// ---- wire mode 1 on the RCCL / host-callback path: y -> bf16, all-reduce(bf16), resid += f32(sum)
__global__ void __launch_bounds__(256) to_bf16_kernel(uint16_t* __restrict__ o, const float* __restrict__ y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) o[i] = f32_to_bf16(y[i]);
}
__global__ void __launch_bounds__(256) add_bf16_kernel(float* __restrict__ resid, const uint16_t* __restrict__ s, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) resid[i] += bf16_to_f32(s[i]);
}

__global__ void __launch_bounds__(256) add_f32_kernel(float* __restrict__ resid, const float* __restrict__ y, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) resid[i] += y[i];
}

// ---- one-shot peer-to-peer all-reduce.  One workgroup owns MI355_P2P_WG_ELEMS elements: publish own slice (write-through
// system-scope stores), drain, raise flag[parity][wg] = seq; wait for every peer's flag; sum the W slices in RANK ORDER.
// Two parities suffice: a rank can only finish call n+1 after every peer has posted n+1, which a peer does after it has
// finished reading call n -- so nobody still reads the slot that call n+2 overwrites.
struct OneShotArgs {
    P2PRegion* local;
    P2PRegion* peer[MI355_P2P_MAX_WORLD];
    int rank, world, mode;                       // mode 0: out = sum ; 1: out = resid + bf16(sum of bf16(y_r))
    const float* y;
    float* out;
    const float* resid;
    int64_t count;
};
#define P2P_SPIN_LIMIT (1 << 22)
__global__ void __launch_bounds__(1024) oneshot_allreduce_kernel(const OneShotArgs a) {
    __shared__ uint32_t s_seq;
    __shared__ int s_lost;
    const int g = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) {
        s_lost = 0;
        // the call counter lives in this rank's own region: a captured graph replays with fresh sequence numbers
        const uint32_t s = __hip_atomic_load(&a.local->ctr[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) + 1u;
        __hip_atomic_store(&a.local->ctr[g], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        s_seq = s;
    }
    __syncthreads();
    const uint32_t seq = s_seq;
    const int par = (int)(seq & 1u);
    const int64_t base = (int64_t)g * MI355_P2P_WG_ELEMS;
    constexpr int PER = MI355_P2P_WG_ELEMS / 1024;
    float v[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int64_t i = base + tid + 1024 * j;
        v[j] = 0.f;
        if (i < a.count) {
            float x = a.y[i];
            if (a.mode == 1) x = bf16_round_f(x);
            v[j] = x;
            __hip_atomic_store(&a.local->stage[par][i], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // this wave's write-through stores have left
    __syncthreads();
    if (tid == 0) __hip_atomic_store(&a.local->flag[par][g], seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (tid < a.world && tid != a.rank) {
        int spins = 0;
        while ((int32_t)(__hip_atomic_load(&a.peer[tid]->flag[par][g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - seq) < 0) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > P2P_SPIN_LIMIT) {                           // bounded: a lost peer is an error, never a hang
                __hip_atomic_store(&a.local->err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                s_lost = 1;
                break;
            }
        }
    }
    __syncthreads();
    // a peer that missed the spin bound makes this call's sum meaningless: poison the outputs (NaN travels to the logits and
    // the sampled token of every rank that was waiting) instead of adding stale slices -- never a plausible wrong number; the
    // sticky error word is read by mi355_llama_read_tokens / mi355_comm_p2p_error
    const bool lost = s_lost != 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int64_t i = base + tid + 1024 * j;
        if (i >= a.count) continue;
        if (lost) { a.out[i] = __uint_as_float(0x7FC00000u); continue; }
        float acc = 0.f;
        for (int r = 0; r < a.world; ++r)
            acc += (r == a.rank) ? v[j] : __hip_atomic_load(&a.peer[r]->stage[par][i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        a.out[i] = (a.mode == 1) ? a.resid[i] + bf16_round_f(acc) : acc;
    }
}

int oneshot(Comm* c, const float* y, float* out, const float* resid, int64_t count, int mode, hipStream_t st) {
    OneShotArgs a{};
    a.local = c->local;
    for (int r = 0; r < c->world; ++r) a.peer[r] = c->peer[r];
    a.rank = c->rank; a.world = c->world; a.mode = mode;
    a.y = y; a.out = out; a.resid = resid; a.count = count;
    const int wgs = (int)((count + MI355_P2P_WG_ELEMS - 1) / MI355_P2P_WG_ELEMS);
    hipLaunchKernelGGL(oneshot_allreduce_kernel, dim3(wgs), dim3(1024), 0, st, a);
    return (int)hipGetLastError();
}
}  // namespace

int comm_unique_id(void* out128) {
    if (!out128) return (int)hipErrorInvalidValue;
    if (!rccl_load()) return (int)hipErrorSharedObjectInitFailed;
    return g_rccl.get_id(static_cast<NcclId*>(out128)) == 0 ? 0 : (int)hipErrorUnknown;
}

int comm_all_reduce(Comm* c, void* buf, int64_t count, int dtype, int64_t stream) {
    if (!c) return (int)hipErrorNotInitialized;
    if (count <= 0) return 0;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (c->p2p && dtype == MI355_DTYPE_F32 && count * 4 <= MI355_P2P_MAX_BYTES)
        return oneshot(c, static_cast<const float*>(buf), static_cast<float*>(buf), nullptr, count, 0, st);
    if (c->ar) return c->ar(c->user, buf, count, dtype, stream);
    const int dt = nccl_dtype_of(dtype);
    if (!c->nccl || dt < 0) return (int)hipErrorInvalidValue;
    return on_side_stream(c, st, [&](hipStream_t s) {
        return g_rccl.all_reduce(buf, buf, (size_t)count, dt, NCCL_SUM, c->nccl, s) == 0 ? 0 : (int)hipErrorUnknown;
    });
}

int comm_all_reduce_f32(Comm* c, float* y, float* resid, int64_t count, int64_t stream) {
    if (!c) return (int)hipErrorNotInitialized;
    if (count <= 0) return 0;
    if (!resid) return comm_all_reduce(c, y, count, MI355_DTYPE_F32, stream);
    // reference numerics: bf16(partial) on the wire, bf16 sum, back to f32, then the residual (attention.rs:1003-1008)
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (c->p2p && count * 4 <= MI355_P2P_MAX_BYTES) return oneshot(c, y, resid, resid, count, 1, st);
    void* t16 = nullptr;
    const int src = mi355_scratch_get(&t16, MI355_SCR_COMM, (size_t)count * 2, st, false);
    if (src) return src;
    const unsigned blocks = (unsigned)((count + 255) / 256 < 4096 ? (count + 255) / 256 : 4096);
    hipLaunchKernelGGL(to_bf16_kernel, dim3(blocks), dim3(256), 0, st, static_cast<uint16_t*>(t16), y, count);
    const int rc = comm_all_reduce(c, t16, count, MI355_DTYPE_BF16, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(add_bf16_kernel, dim3(blocks), dim3(256), 0, st, resid, static_cast<const uint16_t*>(t16), count);
    return (int)hipGetLastError();
}

int comm_all_gather(Comm* c, const void* send, void* recv, int64_t count, int dtype, int64_t stream) {
    if (!c) return (int)hipErrorNotInitialized;
    if (c->ag) return c->ag(c->user, send, recv, count, dtype, stream);
    const int dt = nccl_dtype_of(dtype);
    if (!c->nccl || dt < 0) return (int)hipErrorInvalidValue;
    return on_side_stream(c, reinterpret_cast<hipStream_t>(stream), [&](hipStream_t s) {
        return g_rccl.all_gather(send, recv, (size_t)count, dt, c->nccl, s) == 0 ? 0 : (int)hipErrorUnknown;
    });
}

// ---- C ABI --------------------------------------------------------------------------------------------------------
extern "C" int mi355_comm_unique_id(void* out128) { return comm_unique_id(out128); }

extern "C" void* mi355_comm_create(const void* id128, int32_t rank, int32_t world) {
    if (!id128 || world < 1 || rank < 0 || rank >= world || !rccl_load()) return nullptr;
    NcclId id;
    memcpy(&id, id128, sizeof(id));
    void* nccl = nullptr;
    if (g_rccl.init_rank(&nccl, world, id, rank) != 0) return nullptr;
    Comm* c = new Comm();
    c->nccl = nccl; c->rank = rank; c->world = world;
    return c;
}
extern "C" void* mi355_comm_create_external(mi355_allreduce_fn all_reduce, mi355_allgather_fn all_gather, void* user) {
    if (!all_reduce || !all_gather) return nullptr;
    Comm* c = new Comm();
    c->ar = all_reduce; c->ag = all_gather; c->user = user;
    return c;
}
extern "C" void mi355_comm_destroy(void* comm) {
    Comm* c = static_cast<Comm*>(comm);
    if (!c) return;
    if (c->nccl && g_rccl.destroy) (void)g_rccl.destroy(c->nccl);
    for (int r = 0; r < MI355_P2P_MAX_WORLD; ++r)
        if (c->peer[r] && c->peer[r] != c->local) (void)hipIpcCloseMemHandle(c->peer[r]);
    if (c->local) (void)hipFree(c->local);
    if (c->ev_in) (void)hipEventDestroy(c->ev_in);
    if (c->ev_out) (void)hipEventDestroy(c->ev_out);
    if (c->cs) (void)hipStreamDestroy(c->cs);
    delete c;
}
extern "C" int mi355_comm_set_options(void* comm, int32_t side_stream, int32_t wire_bf16) {
    Comm* c = static_cast<Comm*>(comm);
    if (!c || wire_bf16 < 0 || wire_bf16 > 1) return (int)hipErrorInvalidValue;
    c->side = side_stream != 0;
    c->wire_bf16 = wire_bf16;
    return 0;
}
extern "C" int mi355_comm_wire_bf16(void* comm) { return comm ? static_cast<Comm*>(comm)->wire_bf16 : 0; }

// one-shot path, step 1: allocate this rank's region and export it (64-byte IPC handle the launcher ships to every rank)
extern "C" int mi355_comm_p2p_export(void* comm, void* handle_out64) {
    Comm* c = static_cast<Comm*>(comm);
    if (!c || !handle_out64) return (int)hipErrorInvalidValue;
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
    if (!c->local) {
        void* p = nullptr;
        // fine-grained (system-coherent) memory: peers poll the flags and read the slices while the kernel runs
        // (no coarse-grained fallback: on plain hipMalloc memory the system-scope flag / slice protocol is not guaranteed to be
        // visible to a peer while the kernel runs -- the export fails and the caller keeps RCCL)
        hipError_t e = hipExtMallocWithFlags(&p, sizeof(P2PRegion), hipDeviceMallocFinegrained);
        if (e != hipSuccess) { (void)hipGetLastError(); return (int)e; }
        e = hipMemset(p, 0, sizeof(P2PRegion));
        if (e != hipSuccess) { (void)hipFree(p); return (int)e; }
        CCHECK(hipDeviceSynchronize());
        c->local = static_cast<P2PRegion*>(p);
    }
    hipIpcMemHandle_t h;
    CCHECK(hipIpcGetMemHandle(&h, c->local));
    memcpy(handle_out64, &h, 64);
    return 0;
}
// step 2: open every peer's region (handles = world x 64 bytes in rank order; this rank's own entry is not opened)
extern "C" int mi355_comm_p2p_attach(void* comm, const void* handles, int32_t rank, int32_t world) {
    Comm* c = static_cast<Comm*>(comm);
    if (!c || !handles || !c->local || world < 1 || world > MI355_P2P_MAX_WORLD || rank < 0 || rank >= world) return (int)hipErrorInvalidValue;
    for (int r = 0; r < world; ++r) {
        if (r == rank) { c->peer[r] = c->local; continue; }
        hipIpcMemHandle_t h;
        memcpy(&h, static_cast<const char*>(handles) + 64 * r, 64);
        void* p = nullptr;
        CCHECK(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
        c->peer[r] = static_cast<P2PRegion*>(p);
    }
    c->rank = rank; c->world = world; c->p2p = true;
    return 0;
}
// the launcher's switch after a self-test across the ranks (candle_vllm_amd/model.py::init_comm): 0 = keep the regions but route every
// all-reduce through the communicator's own transport again; 1 = back on (only when every peer region is open).  All ranks must
// make the same call -- a rank that alone leaves the peer kernel would never publish its slice.
extern "C" int mi355_comm_p2p_enable(void* comm, int32_t on) {
    Comm* c = static_cast<Comm*>(comm);
    if (!c) return (int)hipErrorInvalidValue;
    if (!on) { c->p2p = false; return 0; }
    if (!c->local || c->world < 1 || c->world > MI355_P2P_MAX_WORLD) return (int)hipErrorInvalidValue;
    for (int r = 0; r < c->world; ++r) if (!c->peer[r]) return (int)hipErrorInvalidValue;
    c->p2p = true;
    return 0;
}
// 0 = every wait so far met its peer; 1 = a peer did not arrive within the spin bound (results of that call are invalid)
extern "C" int mi355_comm_p2p_error(void* comm) {
    Comm* c = static_cast<Comm*>(comm);
    if (!c || !c->local) return 0;
    uint32_t e = 0;
    if (hipMemcpy(&e, &c->local->err, 4, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return (int)e;
}

// Can this stack capture EVERY collective flavour a TP step issues in a hipGraph?  Captures, on `stream`, a small f32 all-reduce
// (the one-shot peer kernel when attached), a hidden-sized f32 all-reduce beyond the one-shot limit, a bf16 all-reduce (the
// reference's wire numerics) and an all-gather (the vocab-parallel logits), instantiates the graph and destroys it WITHOUT
// launching: nothing runs on the wire, so a rank can test locally and the ranks agree on graph / eager steps before the first real
// step (a per-rank fallback after a failed first step would leave its peers inside a collective).  A communicator with
// ANY host-supplied callback is refused outright: the logits all-gather (and every all-reduce beyond the one-shot limit) would be
// a host call made during capture -- not replayed, i.e. stale logits on every replay.  0 = capturable.
extern "C" int mi355_comm_capture_probe(void* comm, int64_t stream) {
    Comm* c = static_cast<Comm*>(comm);
    if (!c || stream == 0) return (int)hipErrorInvalidValue;
    if (c->ar || c->ag) return (int)hipErrorNotSupported;         // host-supplied collectives are host calls: never captured
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int world = c->world > 0 ? c->world : 1;
    const int64_t big = MI355_P2P_MAX_BYTES / 4 + 1024;           // f32 elements: past the one-shot kernel's limit
    const size_t bytes = (size_t)big * sizeof(float) + (size_t)big * 2 + 256 * sizeof(float) * (size_t)(world + 1);
    char* buf = nullptr;
    CCHECK(hipMalloc(reinterpret_cast<void**>(&buf), bytes));
    (void)hipMemsetAsync(buf, 0, bytes, st);
    (void)hipStreamSynchronize(st);
    float* f_big = reinterpret_cast<float*>(buf);
    uint16_t* h_big = reinterpret_cast<uint16_t*>(buf + (size_t)big * sizeof(float));
    float* g_send = reinterpret_cast<float*>(buf + (size_t)big * sizeof(float) + (size_t)big * 2);
    float* g_recv = g_send + 256;
    int rc = (int)hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    hipGraph_t g = nullptr;
    if (rc == 0) {
        rc = comm_all_reduce(c, f_big, 64, MI355_DTYPE_F32, stream);
        if (rc == 0) rc = comm_all_reduce(c, f_big, big, MI355_DTYPE_F32, stream);
        if (rc == 0) rc = comm_all_reduce(c, h_big, big, MI355_DTYPE_BF16, stream);
        if (rc == 0) rc = comm_all_gather(c, g_send, g_recv, 256, MI355_DTYPE_F32, stream);
        const int erc = (int)hipStreamEndCapture(st, &g);         // always close the capture, also after a refused enqueue
        if (rc == 0) rc = erc;
    }
    if (rc == 0 && g) {
        hipGraphExec_t ge = nullptr;
        rc = (int)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        if (ge) (void)hipGraphExecDestroy(ge);
    }
    if (g) (void)hipGraphDestroy(g);
    if (rc) (void)hipGetLastError();
    (void)hipFree(buf);
    return rc;
}

extern "C" int mi355_comm_all_reduce(void* comm, void* buf, int64_t count, int32_t dtype, int64_t stream) {
    if (!comm || nccl_dtype_of(dtype) < 0) return (int)hipErrorInvalidValue;
    return comm_all_reduce(static_cast<Comm*>(comm), buf, count, dtype, stream);
}
/* resid += sum over ranks of y, with the numerics the communicator is set to (mi355_comm_set_options): wire 0 = f32 sum,
 * wire 1 = the reference's bf16 wire (attention.rs:1003-1008) */
extern "C" int mi355_comm_all_reduce_residual(void* comm, float* y, float* resid, int64_t count, int64_t stream) {
    Comm* c = static_cast<Comm*>(comm);
    if (!c || !y || !resid) return (int)hipErrorInvalidValue;
    if (c->wire_bf16) return comm_all_reduce_f32(c, y, resid, count, stream);
    const int rc = comm_all_reduce(c, y, count, MI355_DTYPE_F32, stream);
    if (rc) return rc;
    const unsigned blocks = (unsigned)((count + 255) / 256 < 4096 ? (count + 255) / 256 : 4096);
    hipLaunchKernelGGL(add_f32_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), resid, y, count);
    return (int)hipGetLastError();
}
extern "C" int mi355_comm_all_gather(void* comm, const void* send, void* recv, int64_t count, int32_t dtype, int64_t stream) {
    if (!comm || nccl_dtype_of(dtype) < 0) return (int)hipErrorInvalidValue;
    return comm_all_gather(static_cast<Comm*>(comm), send, recv, count, dtype, stream);
}
