// Mixture-of-experts routing for the GGUF llama path (Mixtral): `MlpOrMoe::forward`, src/openai/models/quantized_llama.rs:
// 56-123.  The reference pulls the routing weights to the host (`to_vec2`, :70), sorts there and issues per-expert
// index_select / index_add; here routing stays on the device so the decode graph has no host round trip:
//   mi355_moe_route   : rms_norm -> router logits -> softmax -> top-k -> renormalised weights     (one workgroup / token)
//   expert mat-vecs   : qmm_kernel with blockIdx.y = (token, slot) pair and the expert id read on the device
//   mi355_moe_combine : residual += sum_j w_j * expert_j(x)
#include "common.h"
#include "../../include/mi355_vllm.h"

__global__ void __launch_bounds__(256) moe_route_kernel(int32_t* __restrict__ ids, float* __restrict__ wts, const float* __restrict__ x,
                                                        const float* __restrict__ norm_w, float eps, const float* __restrict__ gate,
                                                        int hidden, int E, int K) {
    __shared__ float red[16];
    __shared__ float s_logit[256];
    const int t = blockIdx.x;
    const float* xr = x + (size_t)t * hidden;
    float inv = 1.f;
    if (norm_w) {
        float ss = 0.f;
        for (int i = threadIdx.x; i < hidden; i += blockDim.x) ss += xr[i] * xr[i];
        inv = rsqrtf(block_sum(ss, red) / (float)hidden + eps);
    }
    // router logits: one WAVE per expert (e = wave, wave + 4, ...), 16-byte loads, all of a lane's loads independent --
    // the first version walked the experts one after the other through block-wide reductions (97 us per launch at
    // Mixtral's 8 x 4096: a chain of dependent load latencies)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int e = wave; e < E; e += 4) {
        const float* g = gate + (size_t)e * hidden;
        float acc = 0.f;
        if ((hidden & 3) == 0) {
            for (int i = 4 * lane; i < hidden; i += 256) {
                const float4 xv = *reinterpret_cast<const float4*>(xr + i), gv = *reinterpret_cast<const float4*>(g + i);
                float4 nv = make_float4(1.f, 1.f, 1.f, 1.f);
                if (norm_w) nv = *reinterpret_cast<const float4*>(norm_w + i);
                acc = fmaf(xv.x * inv * nv.x, gv.x, fmaf(xv.y * inv * nv.y, gv.y, fmaf(xv.z * inv * nv.z, gv.z, fmaf(xv.w * inv * nv.w, gv.w, acc))));
            }
        } else {
            for (int i = lane; i < hidden; i += 64) acc = fmaf(norm_w ? xr[i] * inv * norm_w[i] : xr[i], g[i], acc);
        }
        acc = wave_sum(acc);
        if (lane == 0) s_logit[e] = acc;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float m = -INFINITY;
        for (int e = 0; e < E; ++e) m = fmaxf(m, s_logit[e]);
        float den = 0.f;
        for (int e = 0; e < E; ++e) { s_logit[e] = expf(s_logit[e] - m); den += s_logit[e]; }
        for (int e = 0; e < E; ++e) s_logit[e] /= den;          // softmax_last_dim
        float sum = 0.f;
        for (int j = 0; j < K; ++j) {                           // selection sort = stable descending order
            int best = -1; float bv = -INFINITY;
            for (int e = 0; e < E; ++e) {
                bool taken = false;
                for (int q = 0; q < j; ++q) taken |= ids[(size_t)t * K + q] == e;
                if (!taken && s_logit[e] > bv) { bv = s_logit[e]; best = e; }
            }
            ids[(size_t)t * K + j] = best;
            wts[(size_t)t * K + j] = bv;
            sum += bv;
        }
        for (int j = 0; j < K; ++j) wts[(size_t)t * K + j] /= sum;
    }
}

__global__ void __launch_bounds__(256) moe_combine_kernel(float* __restrict__ ys, const float* __restrict__ yp, const float* __restrict__ wts,
                                                          int hidden, int K, int accumulate) {
    const int t = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= hidden) return;
    float acc = accumulate ? ys[(size_t)t * hidden + i] : 0.f;
    for (int j = 0; j < K; ++j) acc = fmaf(wts[(size_t)t * K + j], yp[((size_t)t * K + j) * hidden + i], acc);
    ys[(size_t)t * hidden + i] = acc;
}

// ---- grouped experts (prompt steps / large batches): the (token, slot) pairs sorted by expert, so that every selected
// expert is streamed ONCE per step over all of its tokens -- what the reference's host-routed loop does per expert with
// index_select / index_add (quantized_llama.rs:93-119; layers/moe.rs:746-810)
__global__ void __launch_bounds__(256) moe_gather_kernel(float* __restrict__ dst, const float* __restrict__ src, const int32_t* __restrict__ perm,
                                                         int K, int hidden) {
    const int p = blockIdx.y;                                          // position in expert order
    const int t = perm[p] / K;                                         // pair index = token * K + slot
    const int i = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i < hidden) *reinterpret_cast<float4*>(dst + (size_t)p * hidden + i) = *reinterpret_cast<const float4*>(src + (size_t)t * hidden + i);
}
__global__ void __launch_bounds__(256) moe_scatter_combine_kernel(float* __restrict__ ys, const float* __restrict__ yg, const float* __restrict__ wts,
                                                                  const int32_t* __restrict__ inv, int hidden, int K) {
    const int t = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= hidden) return;
    float acc = ys[(size_t)t * hidden + i];                            // residual (quantized_llama.rs:470)
    for (int j = 0; j < K; ++j) acc = fmaf(wts[(size_t)t * K + j], yg[(size_t)inv[t * K + j] * hidden + i], acc);
    ys[(size_t)t * hidden + i] = acc;
}
// ---- the same grouping decided ON THE DEVICE (decode steps inside a captured graph: no host round trip, fixed launch shapes).
// Every expert owns `cap` rows of the gathered buffers (cap >= the step's pair count: any routing fits); pair p lands in row
// pos[p] = ids[p] * cap + (number of earlier pairs of the same expert) -- stable, token order inside an expert.  Rows past an
// expert's count are never written and never read back: the expert GEMMs run over all cap rows (rows are independent), the
// combine reads only pos[].
__global__ void __launch_bounds__(256) moe_group_kernel(int32_t* __restrict__ pos, int32_t* __restrict__ counts, const int32_t* __restrict__ ids,
                                                        int pairs, int n_expert, int cap, int row_limit) {
    if (counts)
        for (int e = threadIdx.x; e < n_expert; e += blockDim.x) {
            int n = 0;
            for (int q = 0; q < pairs; ++q) n += ids[q] == e ? 1 : 0;
            counts[e] = n;
        }
    for (int p = threadIdx.x; p < pairs; p += blockDim.x) {
        const int e = ids[p];
        int before = 0;
        for (int q = 0; q < p; ++q) before += ids[q] == e ? 1 : 0;
        // an id outside [0, n_expert) (a corrupted router output) gets the DUMP row n_expert * cap: no expert's block, never read by an
        // expert GEMM, so it cannot collide with expert 0's first pair; callers size their row buffers (n_expert * cap + 1)
        // a rank at or beyond `row_limit` (the rows the caller's launches cover per expert: only possible when a token carries the same
        // expert twice, i.e. a corrupted router output) goes to the DUMP row as well -- the host layer keeps that row NaN in the expert
        // outputs, so such a step fails loudly instead of combining rows nobody computed
        pos[p] = (e >= 0 && e < n_expert && before < row_limit) ? e * cap + before : n_expert * cap;
    }
}
__global__ void __launch_bounds__(256) moe_gather_pos_kernel(float* __restrict__ dst, const float* __restrict__ src, const int32_t* __restrict__ pos,
                                                             int K, int hidden) {
    const int p = blockIdx.y;                                          // pair index = token * K + slot
    const int i = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i < hidden) *reinterpret_cast<float4*>(dst + (size_t)pos[p] * hidden + i) = *reinterpret_cast<const float4*>(src + (size_t)(p / K) * hidden + i);
}
/* internal (host_model.cpp): mi355_moe_group with the number of rows per expert that the caller's launches will actually compute */
extern "C" int mi355_internal_moe_group_limited(int32_t* pos, int32_t* counts, const int32_t* expert_ids, int32_t num_pairs, int32_t n_expert,
                                                int32_t cap, int32_t row_limit, int64_t stream) {
    if (num_pairs <= 0) return 0;
    if (!pos || !expert_ids || n_expert < 1 || cap < num_pairs || num_pairs > 4096 || row_limit < 1) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(moe_group_kernel, dim3(1), dim3(256), 0, to_stream(stream), pos, counts, expert_ids, num_pairs, n_expert, cap, row_limit);
    return (int)hipGetLastError();
}
extern "C" int mi355_moe_group(int32_t* pos, int32_t* counts, const int32_t* expert_ids, int32_t num_pairs, int32_t n_expert, int32_t cap,
                               int64_t stream) {
    return mi355_internal_moe_group_limited(pos, counts, expert_ids, num_pairs, n_expert, cap, cap, stream);
}
// ---- grouping for PROMPT steps, on the device (round 6; the reference sorts the pairs on the host after `to_vec2`, quantized_llama.rs:70-91,
// and rounds 2-5 did the same behind one stream synchronisation per layer).  Every expert owns whole 64-row blocks of the gathered buffers --
// expert e's rows start at row 64 * (blocks of the experts before it) -- so a prompt GEMM workgroup (64 token rows) never sees two experts,
// and ONE launch walks the blocks of all experts: block b of `block_table` = {expert, one past the expert's last row}, {0, 0} for the blocks
// past the last expert's (such a workgroup leaves at once).  pos[p] = the row of pair p: stable (token order inside an expert), as the host
// sort was.  One workgroup: counts, then ranks, chunk by chunk from ballots (1024 pairs per trip).
__global__ void __launch_bounds__(1024) moe_group_blocks_kernel(int32_t* __restrict__ pos, int32_t* __restrict__ block_table,
                                                                const int32_t* __restrict__ ids, int num_pairs, int E, int n_blocks) {
    __shared__ int cnt[16], run[16], blk0[17];
    __shared__ int wcnt[16][16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 16) { cnt[tid] = 0; run[tid] = 0; }
    __syncthreads();
    {   // counts: ballots, lane q of every wave keeps expert q's (one LDS atomic per thread serialises 64 lanes on 8 words: 0.7 ms per call)
        int mine = 0;
        for (int c0 = 0; c0 < num_pairs; c0 += 1024) {
            const int p = c0 + tid;
            int e = p < num_pairs ? ids[p] : -1;
            if (p < num_pairs) e = e < 0 ? 0 : (e >= E ? E - 1 : e);
            for (int q = 0; q < E; ++q) {
                const unsigned long long m = __ballot(e == q);
                if (lane == q) mine += __popcll(m);
            }
        }
        if (lane < E) atomicAdd(&cnt[lane], mine);
    }
    __syncthreads();
    if (tid == 0) {
        blk0[0] = 0;
        for (int e = 0; e < E; ++e) blk0[e + 1] = blk0[e] + (cnt[e] + 63) / 64;
    }
    __syncthreads();
    for (int b = tid; b < n_blocks; b += 1024) {
        int e = -1;
        for (int q = 0; q < E; ++q) if (b >= blk0[q] && b < blk0[q + 1]) e = q;
        block_table[2 * b] = e < 0 ? 0 : e;
        block_table[2 * b + 1] = e < 0 ? 0 : 64 * blk0[e] + cnt[e];
    }
    for (int c0 = 0; c0 < num_pairs; c0 += 1024) {
        const int p = c0 + tid;
        int e = p < num_pairs ? ids[p] : -1;
        if (p < num_pairs) e = e < 0 ? 0 : (e >= E ? E - 1 : e);
        int below = 0;                                                  // pairs of my expert in lower lanes of this wave
        for (int q = 0; q < E; ++q) {
            const unsigned long long m = __ballot(e == q);
            if (lane == 0) wcnt[wave][q] = __popcll(m);
            if (e == q) below = __popcll(m & ((1ull << lane) - 1ull));
        }
        __syncthreads();
        if (e >= 0) {
            int r = run[e] + below;
            for (int w = 0; w < wave; ++w) r += wcnt[w][e];
            pos[p] = 64 * blk0[e] + r;
        }
        __syncthreads();
        if (tid < E) {
            int t = 0;
            for (int w = 0; w < 16; ++w) t += wcnt[w][tid];
            run[tid] += t;
        }
        __syncthreads();
    }
}
extern "C" int mi355_moe_group_blocks(int32_t* pos, int32_t* block_table, const int32_t* expert_ids, int32_t num_pairs, int32_t n_expert,
                                      int32_t n_blocks, int64_t stream) {
    if (num_pairs <= 0) return 0;
    if (!pos || !block_table || !expert_ids || n_expert < 1 || n_expert > 16 || n_blocks < (num_pairs + 63) / 64 + n_expert)
        return (int)hipErrorInvalidValue;                               // n_blocks: at least ceil(pairs / 64) + n_expert (every expert may end on a partial block)
    hipLaunchKernelGGL(moe_group_blocks_kernel, dim3(1), dim3(1024), 0, to_stream(stream), pos, block_table, expert_ids, num_pairs, n_expert, n_blocks);
    return (int)hipGetLastError();
}
extern "C" int mi355_moe_gather_pos(float* dst, const float* src, const int32_t* pos, int32_t num_pairs, int32_t top_k, int32_t hidden,
                                    int64_t stream) {
    if (num_pairs <= 0) return 0;
    if (!dst || !src || !pos || top_k < 1 || hidden <= 0 || (hidden & 3)) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(moe_gather_pos_kernel, dim3((hidden / 4 + 255) / 256, num_pairs), dim3(256), 0, to_stream(stream), dst, src, pos, top_k, hidden);
    return (int)hipGetLastError();
}
extern "C" int mi355_moe_gather(float* dst, const float* src, const int32_t* perm, int32_t num_pairs, int32_t top_k, int32_t hidden,
                                int64_t stream) {
    if (num_pairs <= 0) return 0;
    if (!dst || !src || !perm || top_k < 1 || hidden <= 0 || (hidden & 3)) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(moe_gather_kernel, dim3((hidden / 4 + 255) / 256, num_pairs), dim3(256), 0, to_stream(stream), dst, src, perm, top_k, hidden);
    return (int)hipGetLastError();
}
extern "C" int mi355_moe_scatter_combine(float* ys, const float* y_sorted, const float* weights, const int32_t* inv, int32_t num_tokens,
                                         int32_t hidden, int32_t top_k, int64_t stream) {
    if (num_tokens <= 0) return 0;
    if (!ys || !y_sorted || !weights || !inv || top_k < 1 || hidden <= 0) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(moe_scatter_combine_kernel, dim3((hidden + 255) / 256, num_tokens), dim3(256), 0, to_stream(stream), ys, y_sorted,
                       weights, inv, hidden, top_k);
    return (int)hipGetLastError();
}

// Router for 1..16 experts (a power of two) and hidden a multiple of 1024: 16 waves, 16 / E of them per expert, and EVERY
// load of the launch -- x for the sum of squares, x / RMSNorm weight / router row for the dot products -- is requested before
// the first dependent instruction (eight 16-byte pieces per array in flight per lane).  The 256-thread kernel above walks 16
// dependent iterations per expert: 26 us per launch at Mixtral's 8 x 4096, 12 % of its decode step.
__global__ void __launch_bounds__(1024) moe_route16_kernel(int32_t* __restrict__ ids, float* __restrict__ wts, const float* __restrict__ x,
                                                           const float* __restrict__ norm_w, float eps, const float* __restrict__ gate,
                                                           int hidden, int E, int K) {
    __shared__ float red[32];
    __shared__ float s_part[16];
    __shared__ __attribute__((aligned(16))) float xs[8192];          // the normalised row x * inv * w, built once (round 6)
    const int t = blockIdx.x;
    const float* xr = x + (size_t)t * hidden;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wpe = 16 / E, e = wave / wpe, part = wave % wpe;       // waves per expert, this wave's expert and slice of hidden
    const int span = hidden / wpe, base = part * span;
    const float* g = gate + (size_t)e * hidden;
    float acc = 0.f, ss = 0.f;
    // thread i owns elements 4i .. 4i+3 (+ 4096 j) of the row: sum of squares, then the normalised values into LDS.  Rounds 3-5 had every
    // one of the 16 waves re-read its slice of x and of the RMSNorm weight from global memory: 390 KB through ONE CU's vector memory path
    // (64 B per clock: 2.6 us of the launch's 9.9); now x and the weight cross it once (32 KB) beside the 128 KB of router rows.
    float4 xq[2], nq[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int i = 4 * (int)threadIdx.x + 4096 * j;
        xq[j] = i < hidden ? *reinterpret_cast<const float4*>(xr + i) : make_float4(0.f, 0.f, 0.f, 0.f);
        nq[j] = (norm_w && i < hidden) ? *reinterpret_cast<const float4*>(norm_w + i) : make_float4(1.f, 1.f, 1.f, 1.f);
    }
    for (int c0 = 0; c0 < span; c0 += 2048) {                         // 8 pieces of 256 elements per lane and pass
        float4 gv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = base + c0 + 256 * u + 4 * lane;
            const bool in = c0 + 256 * u < span;
            gv[u] = in ? *reinterpret_cast<const float4*>(g + i) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (c0 == 0) {
            if (norm_w) {
#pragma unroll
                for (int j = 0; j < 2; ++j) ss += xq[j].x * xq[j].x + xq[j].y * xq[j].y + xq[j].z * xq[j].z + xq[j].w * xq[j].w;
                // (hidden <= 8192: the two pieces above are the whole row); DPP sums: every ds_bpermute of wave_sum is an LDS round trip
                ss = wave_sum_dpp(ss);
                if (lane == 0) red[wave] = ss;
                __syncthreads();
                ss = wave_sum_dpp(lane < 16 ? red[lane] : 0.f);
            }
            const float inv = norm_w ? rsqrtf(ss / (float)hidden + eps) : 1.f;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int i = 4 * (int)threadIdx.x + 4096 * j;
                if (i < hidden)
                    *reinterpret_cast<float4*>(xs + i) = make_float4(xq[j].x * inv * nq[j].x, xq[j].y * inv * nq[j].y, xq[j].z * inv * nq[j].z, xq[j].w * inv * nq[j].w);
            }
            __syncthreads();
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = base + c0 + 256 * u + 4 * lane;
            const bool in = c0 + 256 * u < span;
            const float4 xn = in ? *reinterpret_cast<const float4*>(xs + i) : make_float4(0.f, 0.f, 0.f, 0.f);
            acc = fmaf(xn.x, gv[u].x, fmaf(xn.y, gv[u].y, fmaf(xn.z, gv[u].z, fmaf(xn.w, gv[u].w, acc))));
        }
    }
    acc = wave_sum_dpp(acc);
    if (lane == 0) s_part[wave] = acc;
    __syncthreads();
    // ---- softmax + top-k by wave 0 IN REGISTERS (round 6).  The first form ran this tail on one thread over an LDS array (~60 dependent LDS
    // accesses: max, exp, sum, divide, K selection scans; measured worth 0.5-1 % of the Mixtral batch-1 step).  Here lane q adds expert q's partial dot
    // products, every lane gathers the E logits by readlane, and the scans run on registers -- same order of every sum and the same
    // first-maximum-wins selection as before (bit-identical weights and ids).
    if (wave != 0) return;
    float l = 0.f;
    if (lane < E)
        for (int q = 0; q < wpe; ++q) l += s_part[lane * wpe + q];
    float pr[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) pr[q] = __shfl(l, q, 64);
    float m = -INFINITY;
#pragma unroll
    for (int q = 0; q < 16; ++q) if (q < E) m = fmaxf(m, pr[q]);
    float den = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) if (q < E) { pr[q] = expf(pr[q] - m); den += pr[q]; }
#pragma unroll
    for (int q = 0; q < 16; ++q) if (q < E) pr[q] /= den;          // softmax_last_dim
    float sum = 0.f, wmine = 0.f;
    unsigned taken = 0;
    int imine = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {                          // selection sort = stable descending order; lane j keeps pick j
        if (j >= K) break;
        int best = -1; float bv = -INFINITY;
#pragma unroll
        for (int q = 0; q < 16; ++q)
            if (q < E && !((taken >> q) & 1u) && pr[q] > bv) { bv = pr[q]; best = q; }
        taken |= 1u << best;
        if (lane == j) { imine = best; wmine = bv; }
        sum += bv;
    }
    if (lane < K) {
        ids[(size_t)t * K + lane] = imine;
        wts[(size_t)t * K + lane] = wmine / sum;
    }
}

extern "C" int mi355_moe_route(int32_t* expert_ids, float* weights, const float* x, const float* norm_weight, float norm_eps,
                               const float* gate_inp, int32_t num_tokens, int32_t hidden, int32_t n_expert, int32_t top_k,
                               int64_t stream) {
    if (num_tokens <= 0) return 0;
    if (n_expert < 1 || n_expert > 256 || top_k < 1 || top_k > n_expert) return (int)hipErrorInvalidValue;
    if (n_expert <= 16 && (n_expert & (n_expert - 1)) == 0 && (hidden % 1024) == 0 && hidden <= 8192 &&
        (hidden / (16 / n_expert)) % 256 == 0)
        hipLaunchKernelGGL(moe_route16_kernel, dim3(num_tokens), dim3(1024), 0, to_stream(stream), expert_ids, weights, x,
                           norm_weight, norm_eps, gate_inp, hidden, n_expert, top_k);
    else
        hipLaunchKernelGGL(moe_route_kernel, dim3(num_tokens), dim3(256), 0, to_stream(stream), expert_ids, weights, x, norm_weight,
                           norm_eps, gate_inp, hidden, n_expert, top_k);
    return (int)hipGetLastError();
}
extern "C" int mi355_moe_combine(float* ys, const float* y_pairs, const float* weights, int32_t num_tokens, int32_t hidden,
                                 int32_t top_k, int32_t accumulate, int64_t stream) {
    if (num_tokens <= 0) return 0;
    hipLaunchKernelGGL(moe_combine_kernel, dim3((hidden + 255) / 256, num_tokens), dim3(256), 0, to_stream(stream), ys, y_pairs,
                       weights, hidden, top_k, accumulate);
    return (int)hipGetLastError();
}
