// Tensor-parallel communicator of the host layers (comm.hip): what every mi355_comm_* handle points to.
//   reference: one nccl `Comm` per process (src/openai/pipelines/pipeline.rs:805-812), collectives C1/C2/C3 of
//   src/openai/distributed.rs:547-654,696-711,1335-1446,1632-1667
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mi355_vllm.h"

#define MI355_P2P_MAX_WORLD 8
#define MI355_P2P_MAX_BYTES (256 * 1024)        // one-shot path: decode-sized messages (8-256 KiB)
#define MI355_P2P_WG_ELEMS 4096                 // f32 elements one workgroup of the one-shot kernel owns
#define MI355_P2P_MAX_WG (MI355_P2P_MAX_BYTES / 4 / MI355_P2P_WG_ELEMS)

// layout of one rank's exported region (fine-grained device memory, opened by every peer through an IPC handle)
struct P2PRegion {
    uint32_t flag[2][MI355_P2P_MAX_WG];         // flag[parity][wg] = sequence number of the slice published there
    uint32_t ctr[MI355_P2P_MAX_WG];             // this rank's per-workgroup call counter (lives on the device: graph replay)
    uint32_t err;                               // a peer never arrived within the spin bound
    uint32_t pad[1024 - 3 * MI355_P2P_MAX_WG - 1];
    float stage[2][MI355_P2P_MAX_BYTES / 4];
};

struct Comm {
    void* nccl = nullptr;                       // RCCL communicator (dlopen) ...
    mi355_allreduce_fn ar = nullptr;            // ... or collectives supplied by the host
    mi355_allgather_fn ag = nullptr;
    void* user = nullptr;
    int rank = 0, world = 1;
    // RCCL calls run on a side stream fenced by events (north_star; distributed.rs:547-654 enqueues on the nccl stream)
    bool side = true;
    hipStream_t cs = nullptr;
    hipEvent_t ev_in = nullptr, ev_out = nullptr;
    // 0: f32 on the wire, rank 0 carries the residual into the sum (one rounding fewer than the reference)
    // 1: the reference's numerics: every rank rounds its partial to bf16, all-reduce in bf16, back to f32, + residual
    //    (attention.rs:1003-1008, quantized_llama.rs:38-42)
    int wire_bf16 = 0;
    void* tmp16 = nullptr;                      // bf16 staging of the RCCL path in wire mode 1
    size_t tmp16_bytes = 0;
    // one-shot peer-to-peer all-reduce (<= 256 KiB)
    P2PRegion* local = nullptr;
    P2PRegion* peer[MI355_P2P_MAX_WORLD] = {nullptr};
    bool p2p = false;
};

// sum over ranks of y (f32 [count]); result:
//   resid == nullptr : y <- sum                                  (wire 0: rank 0's partial already holds the residual)
//   resid != nullptr : resid <- resid + sum                      (wire 1: sum of bf16-rounded partials, rounded to bf16)
int comm_all_reduce_f32(Comm* c, float* y, float* resid, int64_t count, int64_t stream);
int comm_all_reduce(Comm* c, void* buf, int64_t count, int dtype, int64_t stream);     // plain in-place sum
int comm_all_gather(Comm* c, const void* send, void* recv, int64_t count, int dtype, int64_t stream);
int comm_unique_id(void* out128);
