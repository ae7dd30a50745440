// K4 -- prefill attention: causal, variable-length (cu_seqlens_q), grouped-query, optionally over a cached prefix
// that lives in the paged KV cache (prefix cache / chunked prefill: context_len = cached + chunk).
// The prefill half of PagedAttention::forward (attention-rs; call sites src/openai/models/layers/attention.rs:707-719,
// 983-995; metadata src/openai/pipelines/inputs.rs:90-230,351-367).
//
// MI355X design: a flash-attention loop with everything on the 16x16x32 MFMA and NO LDS / barriers:
//   S^T = K . Q^T   (A = K rows straight from global: 16 B per lane; B = Q held in registers for the whole loop)
//         -> C layout: lane (query = lane&15) holds keys 4*(lane>>4)+v, i.e. each lane owns ONE query: the online
//            softmax state (m, l) and the O rescale factor are per-lane scalars, P never leaves its lane.
//   O^T = V^T . P^T (B = P packed to 16-bit in registers; A = V^T)
// V^T needs 8 keys at a fixed d per lane while row-major V gives 8 d at a fixed key.  The transpose is done by the
// matrix core itself: MFMA(A = V rows, B = identity slice) returns, in C layout, exactly lane (d, keys 4kg+v) --
// exact in f32, so the round trip through 16-bit is lossless.  The vLLM-style paged V cache [D][block] already has
// keys contiguous and is loaded directly.
// One wave = 32 queries (2 sub-tiles) of one head; a workgroup = 4 waves = 128 consecutive queries (so the K/V
// tiles they share hit L1/L2).  grid = (ceil(max_seqlen_q/128), heads, seqs): >> 256 workgroups for real prompts.
#include "common.h"
#include "pa_lds_layout.h"
#include "scratch.h"
#include <stdlib.h>
#include "../../include/mi355_vllm.h"
#include <hip/hip_runtime.h>

typedef _Float16 pf_f16x8_t __attribute__((ext_vector_type(8)));

enum { SRC_CONTIG = 0, SRC_FLASH = 1, SRC_PAGED = 2, SRC_PAGED8 = 3 };   // PAGED8: e4m3fn cache, K [NB, Hkv, D/16, bs, 16], V [NB, Hkv, D, bs]

struct PrefillParams {
    void* out;                 // [T, H, D]
    const void* q;             // [T, H, D]
    const void* k;             // [T, Hkv, D] (SRC_CONTIG) or key cache
    const void* v;
    const uint32_t* block_tables;   // [num_seqs, max_blocks]
    const uint32_t* context_lens;   // [num_seqs] = cached + chunk, or null (= chunk)
    const uint32_t* cu_q;           // [num_seqs + 1]
    int32_t H, Hkv, block_size, max_blocks;
    float scale_log2, softcap;      // scale * log2(e) (softcap == 0) ; softcap > 0: scale applied before tanh
    float scale;
    int32_t kv8;                    // generic kernel only: e4m3fn cache (PAGED layout, K x = 16)
    float k_scale, v_scale;
    int32_t window;                 // generic kernel only: sliding window (0 = none): the query at position i sees keys i - window + 1 .. i
};

template <int DT>
__device__ __forceinline__ f32x4_t pf_mfma(const uint4& a, const uint4& b, const f32x4_t& c) {
    if constexpr (DT == MI355_DTYPE_BF16)
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(pf_f16x8_t, a), __builtin_bit_cast(pf_f16x8_t, b), c, 0, 0, 0);
}
template <int DT> __device__ __forceinline__ uint32_t pf_pack(float lo, float hi) {
    if constexpr (DT == MI355_DTYPE_BF16) return cvt_pk_bf16(lo, hi);
    else return (uint32_t)f32_to_f16_bits(lo) | ((uint32_t)f32_to_f16_bits(hi) << 16);
}

// 16-byte chunk (8 consecutive d starting at d0) of key/value ROW j of kv head hk
template <int SRC, int D>
__device__ __forceinline__ const uint4* row_chunk(const PrefillParams& p, const void* base, const uint32_t* bt, int q_begin,
                                                  int hk, int j, int d0) {
    const uint16_t* b = static_cast<const uint16_t*>(base);
    if constexpr (SRC == SRC_CONTIG) {
        return reinterpret_cast<const uint4*>(b + ((size_t)(q_begin + j) * p.Hkv + hk) * D + d0);
    } else if constexpr (SRC == SRC_FLASH) {
        const size_t slot = (size_t)bt[j / p.block_size] * p.block_size + j % p.block_size;
        return reinterpret_cast<const uint4*>(b + (slot * p.Hkv + hk) * D + d0);
    } else {   // K cache [NB, Hkv, D/8, bs, 8]
        const size_t blk = bt[j / p.block_size];
        return reinterpret_cast<const uint4*>(b + (((blk * p.Hkv + hk) * (D / 8) + d0 / 8) * p.block_size + j % p.block_size) * 8);
    }
}

// 4 e4m3fn bytes -> 4 bf16 (exact: 3 mantissa bits)
__device__ __forceinline__ uint2 pf_fp8x4_to_bf16x4(uint32_t w) {
    typedef float pf_f32x2 __attribute__((ext_vector_type(2)));
    const pf_f32x2 a = __builtin_amdgcn_cvt_pk_f32_fp8((int)w, false), b = __builtin_amdgcn_cvt_pk_f32_fp8((int)w, true);
    return make_uint2(cvt_pk_bf16(a.x, a.y), cvt_pk_bf16(b.x, b.y));
}
// the 8 values d0 .. d0 + 7 of key j as an MFMA fragment (d0 % 8 == 0)
template <int SRC, int D>
__device__ __forceinline__ uint4 k_frag(const PrefillParams& p, const uint32_t* bt, int q_begin, int hk, int j, int d0) {
    if constexpr (SRC == SRC_PAGED8) {
        const size_t blk = bt[j / p.block_size];
        const uint8_t* b8 = static_cast<const uint8_t*>(p.k) + (((blk * p.Hkv + hk) * (D / 16) + d0 / 16) * p.block_size + j % p.block_size) * 16 + (d0 & 8);
        const uint2 raw = *reinterpret_cast<const uint2*>(b8);
        const uint2 lo = pf_fp8x4_to_bf16x4(raw.x), hi = pf_fp8x4_to_bf16x4(raw.y);
        return make_uint4(lo.x, lo.y, hi.x, hi.y);
    } else {
        return *row_chunk<SRC, D>(p, p.k, bt, q_begin, hk, j, d0);
    }
}

template <int DT, int D, int SRC>
__global__ void __launch_bounds__(256) prefill_attn_kernel(const PrefillParams p) {
    static_assert(SRC != SRC_PAGED8 || DT == MI355_DTYPE_BF16, "e4m3 is exact in bf16 only");
    constexpr int NC = D / 32;     // 32-wide d chunks (QK contraction, V row loads)
    constexpr int NDT = D / 16;    // 16-wide d tiles of O
    const int seq = blockIdx.z, h = blockIdx.y;
    const int q_begin = (int)p.cu_q[seq], qlen = (int)p.cu_q[seq + 1] - q_begin;
    const int ctx = p.context_lens ? (int)p.context_lens[seq] : qlen;
    const int cached = ctx - qlen;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r16 = lane & 15, kg = lane >> 4;
    // query blocks are handed out HEAVIEST FIRST (the last block of a prompt sees every key): the light ones fill in behind them
    const int qw0 = ((int)gridDim.x - 1 - (int)blockIdx.x) * 128 + wave * 32;          // first query (chunk-local) of this wave
    if (qw0 >= qlen) return;
    const int hk = h / (p.H / p.Hkv);
    const uint32_t* bt = p.block_tables ? p.block_tables + (size_t)seq * p.max_blocks : nullptr;
    const uint16_t* q16 = static_cast<const uint16_t*>(p.q);

    // ---- Q fragments (B operand: lane = query r16, k = d 32c + 8kg ..)
    uint4 qf[2][NC];
    int pos[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int ql = qw0 + 16 * s + r16;
        pos[s] = cached + min(ql, qlen - 1);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            qf[s][c] = make_uint4(0, 0, 0, 0);
            if (ql < qlen) qf[s][c] = *reinterpret_cast<const uint4*>(q16 + ((size_t)(q_begin + ql) * p.H + h) * D + 32 * c + 8 * kg);
        }
    }
    // identity slices for the MFMA transpose: B[k][n] = (k == 16*half + n), lane holds k = 8kg .. 8kg+7
    constexpr uint32_t ONE = (DT == MI355_DTYPE_BF16) ? 0x3F80u : 0x3C00u;
    uint4 ident[2];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int i = 16 * half + r16 - 8 * kg;              // element index inside this lane's 8, if 0..7
        uint32_t w[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = (i == 2 * e) ? ONE : ((i == 2 * e + 1) ? (ONE << 16) : 0u);
        ident[half] = make_uint4(w[0], w[1], w[2], w[3]);
    }

    float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};
    f32x4_t o[2][NDT];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) o[s][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const f32x4_t zero4 = {0.f, 0.f, 0.f, 0.f};

    const int kend = min(ctx, cached + min(qw0 + 32, qlen));   // keys this wave can see (exclusive)
    for (int j0 = 0; j0 < kend; j0 += 32) {
        // ---- K rows (A operand: lane = key r16 of key tile kt) and S^T
        uint4 kf[2][NC];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            const int j = min(j0 + 16 * kt + r16, ctx - 1);
#pragma unroll
            for (int c = 0; c < NC; ++c) kf[kt][c] = k_frag<SRC, D>(p, bt, q_begin, hk, j, 32 * c + 8 * kg);
        }
        // ---- V^T fragments (A operand of PV: lane = d r16 of d tile, k = keys {4kg..4kg+3} of kt 0 and of kt 1)
        uint4 vt[NDT];
        if constexpr (SRC == SRC_PAGED8) {
            const uint8_t* vb = static_cast<const uint8_t*>(p.v);
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                uint32_t w[4];
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) {
                    const int jv = j0 + 16 * kt + 4 * kg;                        // 4 consecutive keys, same block: 4 bytes
                    const int jc = min(jv, ctx - 1);
                    const size_t blk = bt[jc / p.block_size];
                    uint32_t raw = *reinterpret_cast<const uint32_t*>(vb + ((blk * p.Hkv + hk) * D + 16 * dt + r16) * p.block_size + (jv % p.block_size));
                    if (jv + 3 >= ctx) {      // bytes past the context: zero them (0x00 = +0.0 in e4m3)
                        if (jv + 0 >= ctx) raw &= 0xFFFFFF00u;
                        if (jv + 1 >= ctx) raw &= 0xFFFF00FFu;
                        if (jv + 2 >= ctx) raw &= 0xFF00FFFFu;
                        if (jv + 3 >= ctx) raw &= 0x00FFFFFFu;
                    }
                    const uint2 t = pf_fp8x4_to_bf16x4(raw);
                    w[2 * kt] = t.x; w[2 * kt + 1] = t.y;
                }
                vt[dt] = make_uint4(w[0], w[1], w[2], w[3]);
            }
        } else if constexpr (SRC == SRC_PAGED) {
            const uint16_t* vb = static_cast<const uint16_t*>(p.v);
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                uint32_t w[4];
#pragma unroll
                for (int kt = 0; kt < 2; ++kt) {
                    const int jv = j0 + 16 * kt + 4 * kg;                        // 4 consecutive keys, same block
                    const int jc = min(jv, ctx - 1);
                    const size_t blk = bt[jc / p.block_size];
                    const uint2 t = *reinterpret_cast<const uint2*>(vb + ((blk * p.Hkv + hk) * D + 16 * dt + r16) * p.block_size + (jv % p.block_size));
                    w[2 * kt] = t.x; w[2 * kt + 1] = t.y;
                    if (jv + 3 >= ctx) {      // slots past the context hold arbitrary bits: zero them (0 * NaN would poison O)
                        if (jv + 0 >= ctx) w[2 * kt] &= 0xFFFF0000u;
                        if (jv + 1 >= ctx) w[2 * kt] &= 0x0000FFFFu;
                        if (jv + 2 >= ctx) w[2 * kt + 1] &= 0xFFFF0000u;
                        if (jv + 3 >= ctx) w[2 * kt + 1] &= 0x0000FFFFu;
                    }
                }
                vt[dt] = make_uint4(w[0], w[1], w[2], w[3]);
            }
        } else {
            uint4 vr[2][NC];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                const int j = min(j0 + 16 * kt + r16, ctx - 1);
#pragma unroll
                for (int c = 0; c < NC; ++c) vr[kt][c] = *row_chunk<SRC == SRC_CONTIG ? SRC_CONTIG : SRC_FLASH, D>(p, p.v, bt, q_begin, hk, j, 32 * c + 8 * kg);
            }
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const f32x4_t t0 = pf_mfma<DT>(vr[0][c], ident[half], zero4);   // lane (d = 32c+16half+r16): keys 4kg+v of kt 0
                    const f32x4_t t1 = pf_mfma<DT>(vr[1][c], ident[half], zero4);
                    vt[2 * c + half] = make_uint4(pf_pack<DT>(t0[0], t0[1]), pf_pack<DT>(t0[2], t0[3]),
                                                  pf_pack<DT>(t1[0], t1[1]), pf_pack<DT>(t1[2], t1[3]));
                }
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            f32x4_t st[2];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                st[kt] = zero4;
#pragma unroll
                for (int c = 0; c < NC; ++c) st[kt] = pf_mfma<DT>(kf[kt][c], qf[s][c], st[kt]);
            }
            // ---- online softmax; this lane owns query r16 of sub-tile s, keys j0 + 16kt + 4kg + v
            float x[8];
            float tmax = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int j = j0 + 16 * kt + 4 * kg + v;
                    float sv = st[kt][v];
                    if (p.softcap > 0.f) sv = tanhf(sv * p.scale / p.softcap) * p.softcap * 1.4426950408889634f;
                    else sv *= p.scale_log2;
                    sv = (j <= pos[s]) ? sv : -INFINITY;          // causal (j < ctx follows from j <= pos)
                    x[4 * kt + v] = sv;
                    tmax = fmaxf(tmax, sv);
                }
            tmax = fmaxf(tmax, __shfl_xor(tmax, 16));
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
            const float m_new = fmaxf(m_run[s], tmax);           // finite: key 0 is visible to every query
            const float alpha = exp2f(m_run[s] - m_new);
            float psum = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) { x[i] = exp2f(x[i] - m_new); psum += x[i]; }
            psum += __shfl_xor(psum, 16);
            psum += __shfl_xor(psum, 32);
            l_run[s] = l_run[s] * alpha + psum;
            m_run[s] = m_new;
            const uint4 pb = make_uint4(pf_pack<DT>(x[0], x[1]), pf_pack<DT>(x[2], x[3]), pf_pack<DT>(x[4], x[5]), pf_pack<DT>(x[6], x[7]));
            // bf16 keeps 8 bits of a probability: carry the rounding residual in a second operand so that P.V
            // matches the f32-softmax definition to ~2^-17 (the 1e-3 logit tolerance is against exact softmax)
            uint4 pl = make_uint4(0, 0, 0, 0);
            if constexpr (DT == MI355_DTYPE_BF16) {
                float r[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const uint32_t w = (i & 1) ? ((&pb.x)[i >> 1] & 0xFFFF0000u) : ((&pb.x)[i >> 1] << 16);
                    r[i] = x[i] - __uint_as_float(w);
                }
                pl = make_uint4(pf_pack<DT>(r[0], r[1]), pf_pack<DT>(r[2], r[3]), pf_pack<DT>(r[4], r[5]), pf_pack<DT>(r[6], r[7]));
            }
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                o[s][dt][0] *= alpha; o[s][dt][1] *= alpha; o[s][dt][2] *= alpha; o[s][dt][3] *= alpha;
                o[s][dt] = pf_mfma<DT>(vt[dt], pb, o[s][dt]);   // C: col = query r16, rows d = 16dt + 4kg + v
                if constexpr (DT == MI355_DTYPE_BF16) o[s][dt] = pf_mfma<DT>(vt[dt], pl, o[s][dt]);
            }
        }
    }
    // ---- epilogue
    uint16_t* out16 = static_cast<uint16_t*>(p.out);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int ql = qw0 + 16 * s + r16;
        if (ql >= qlen) continue;
        const float inv = (SRC == SRC_PAGED8 ? p.v_scale : 1.f) / l_run[s];
        uint16_t* op = out16 + ((size_t)(q_begin + ql) * p.H + h) * D + 4 * kg;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
            uint2 w;
            w.x = pf_pack<DT>(o[s][dt][0] * inv, o[s][dt][1] * inv);
            w.y = pf_pack<DT>(o[s][dt][2] * inv, o[s][dt][3] * inv);
            *reinterpret_cast<uint2*>(op + 16 * dt) = w;
        }
    }
}

// ------------------------------------------------------------------------------------------------ generic fallback
// any head_dim <= 256 / block size: one wave per (query, head); lanes split d, keys visited one by one.
// Correctness path for shapes the MFMA kernel does not cover (e.g. StableLM's D = 80).
template <int DT>
__global__ void __launch_bounds__(64) prefill_attn_generic_kernel(const PrefillParams p, int D, int src) {
    const int seq = blockIdx.z, h = blockIdx.y, ql = blockIdx.x;
    const int q_begin = (int)p.cu_q[seq], qlen = (int)p.cu_q[seq + 1] - q_begin;
    if (ql >= qlen) return;
    const int ctx = p.context_lens ? (int)p.context_lens[seq] : qlen;
    const int cached = ctx - qlen, lane = threadIdx.x;
    const int hk = h / (p.H / p.Hkv), pos = cached + ql;
    const uint32_t* bt = p.block_tables ? p.block_tables + (size_t)seq * p.max_blocks : nullptr;
    const uint16_t* q16 = static_cast<const uint16_t*>(p.q) + ((size_t)(q_begin + ql) * p.H + h) * D;
    auto cvt = [](uint16_t b) { return DT == MI355_DTYPE_BF16 ? bf16_to_f32(b) : f16_bits_to_f32(b); };
    float qv[4], acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 4; ++i) { const int d = lane + 64 * i; qv[i] = d < D ? cvt(q16[d]) : 0.f; }
    float m = -INFINITY, l = 0.f;
    const int j0 = p.window > 0 ? max(0, pos - p.window + 1) : 0;     // sliding window: the last `window` keys (mask.rs:22-27)
    for (int j = j0; j <= pos; ++j) {
        float kv[4], vv[4], dot = 0.f;
        for (int i = 0; i < 4; ++i) {
            const int d = lane + 64 * i;
            kv[i] = vv[i] = 0.f;
            if (d >= D) continue;
            size_t ko, vo;
            if (src == SRC_CONTIG) { ko = vo = ((size_t)(q_begin + j) * p.Hkv + hk) * D + d; }
            else {
                const size_t blk = bt[j / p.block_size]; const int off = j % p.block_size;
                if (src == SRC_FLASH) ko = vo = ((blk * p.block_size + off) * p.Hkv + hk) * D + d;
                else { ko = (((blk * p.Hkv + hk) * (D / 8) + d / 8) * p.block_size + off) * 8 + d % 8;
                       vo = ((blk * p.Hkv + hk) * D + d) * p.block_size + off; }
            }
            if (p.kv8) {
                typedef float pf_f32x2 __attribute__((ext_vector_type(2)));
                const size_t blk = bt[j / p.block_size]; const int off = j % p.block_size;
                ko = (((blk * p.Hkv + hk) * (D / 16) + d / 16) * p.block_size + off) * 16 + d % 16;
                const pf_f32x2 kk = __builtin_amdgcn_cvt_pk_f32_fp8((int)static_cast<const uint8_t*>(p.k)[ko], false);
                const pf_f32x2 vq = __builtin_amdgcn_cvt_pk_f32_fp8((int)static_cast<const uint8_t*>(p.v)[vo], false);
                kv[i] = kk.x * p.k_scale; vv[i] = vq.x * p.v_scale;
            } else {
                kv[i] = cvt(static_cast<const uint16_t*>(p.k)[ko]);
                vv[i] = cvt(static_cast<const uint16_t*>(p.v)[vo]);
            }
            dot += qv[i] * kv[i];
        }
        dot = wave_sum(dot) * p.scale;
        if (p.softcap > 0.f) dot = tanhf(dot / p.softcap) * p.softcap;
        const float m_new = fmaxf(m, dot), alpha = __expf(m - m_new), pj = __expf(dot - m_new);
        l = l * alpha + pj; m = m_new;
        for (int i = 0; i < 4; ++i) acc[i] = acc[i] * alpha + pj * vv[i];
    }
    uint16_t* o16 = static_cast<uint16_t*>(p.out) + ((size_t)(q_begin + ql) * p.H + h) * D;
    for (int i = 0; i < 4; ++i) {
        const int d = lane + 64 * i;
        if (d < D) o16[d] = DT == MI355_DTYPE_BF16 ? f32_to_bf16(acc[i] / l) : f32_to_f16_bits(acc[i] / l);
    }
}

// ------------------------------------------------------------------------------------------------
// EXPERIMENT (tuning key 47 = 1; bf16, head_dim 128, keys from the PAGED cache, block size 16 / 32 / 64): prompt attention with K / V
// through LDS by DMA.  prefill_attn_kernel above gives every wave its own K / V fragments straight from global memory: 128 queries of
// ONE head per workgroup, so a 2048-token prompt reads every K / V tile 16 x 4 (GQA) times out of L2 -- 432 us per layer at
// Llama-3-8B shapes, 79 TFLOP/s.  Here a workgroup is 64 queries x the (up to) four heads of a GQA group: wave w owns head w, the four
// waves share every K / V byte through the stage ring of paged_attn_lds_kernel (64-token stages, global_load_lds_dwordx4 three stages
// ahead, counted vmcnt, one barrier per stage, the same conflict-free layouts -- pa_lds_layout.h), and a wave has the whole register
// file: Q (4 query tiles x 4 k-steps), K of the stage (16 fragments, read once for the four query tiles), O (4 x 8 accumulators).
//   S^T = K . Q^T per (key tile, query tile): the lane owns ONE query per query tile -- running max / sum are per-lane scalars;
//   O   = P . V  with P as the A operand (hi + lo bf16 pieces, as prefill_attn_kernel: the 1e-3 logit bound is against exact
//         softmax) and the V row fragment as B: 192 MFMAs per wave and stage.
// Causal masks only on the stages the diagonal crosses; exp2 domain and softcap as prefill_attn_kernel.
template <int R, int QT = 4>
__global__ void __launch_bounds__(256, 1) prefill_attn_lds_kernel(const PrefillParams p, const uint32_t* __restrict__ btab,
                                                                   const uint32_t* __restrict__ clens, const uint32_t* __restrict__ cuq) {
    constexpr int D = 128, NC = 4, NDT = 8;
    constexpr uint32_t STAGE_B = 32768u;
    extern __shared__ __attribute__((aligned(1024))) uint8_t pfl_smem[];
    // QT = query tiles of 16 per workgroup.  Grid: x = head group (fastest), y = query block, HEAVIEST FIRST (the last block of a prompt sees
    // all its keys, the first one stage): with more workgroups than CUs the light blocks fill in behind the heavy ones
    const int qb = (int)gridDim.y - 1 - (int)blockIdx.y, seq = blockIdx.z;
    const int G = p.H / p.Hkv, HG = (G + 3) >> 2;                     // head groups of up to four query heads per kv head
    const int hk = (int)blockIdx.x / HG, hg = (int)blockIdx.x % HG;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int c = lane & 15, kg = lane >> 4;
    const int bs = p.block_size, bs_shift = __ffs(bs) - 1, E = 64 >> bs_shift;
    const int q_begin = (int)cuq[seq], qlen = (int)cuq[seq + 1] - q_begin;
    const int ctx = (int)clens[seq];
    const int cached = ctx - qlen;
    const int q0 = qb * (16 * QT);
    if (q0 >= qlen) return;                                           // uniform for the workgroup
    const int hw = 4 * hg + wave;
    const bool active = hw < G;                                       // this wave has a head (all waves copy and meet at the barriers)
    const int h = hk * G + (active ? hw : 0);
    const int kend = min(ctx, cached + min(q0 + 16 * QT, qlen));           // keys this block can see (exclusive)
    const int ns = (kend + 63) >> 6;
    const int nblk = (ctx + bs - 1) >> bs_shift;
    // ---- Q fragments (B operand: lane = query c of tile qt, k = d 32j + 8kg ..) and the query's position
    uint4 qf[QT][NC];
    int pos[QT];
    const uint16_t* q16 = static_cast<const uint16_t*>(p.q);
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const int ql = q0 + 16 * qt + c;
        pos[qt] = cached + min(ql, qlen - 1);
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            qf[qt][j] = *reinterpret_cast<const uint4*>(q16 + ((size_t)(q_begin + min(ql, qlen - 1)) * p.H + h) * D + 32 * j + 8 * kg);
            if (ql >= qlen || !active) qf[qt][j] = make_uint4(0, 0, 0, 0);
        }
    }
    // the compiler's wait for these loads must come before the first DMA goes out (it knows nothing of the DMA queue)
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) asm volatile("" : "+v"(qf[qt][0].x), "+v"(qf[qt][1].x), "+v"(qf[qt][2].x), "+v"(qf[qt][3].x));
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(__attribute__((address_space(3))) void*)pfl_smem);
    const uint8_t* kc8 = static_cast<const uint8_t*>(p.k);
    const uint8_t* vc8 = static_cast<const uint8_t*>(p.v);
    const int ktok = pal_dma_token(0, lane);
    auto load_ent = [&](int st, int (&ent)[4]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = min(st * E + min(e, E - 1), nblk - 1);
            ent[e] = (int)btab[(int64_t)seq * p.max_blocks + idx];
        }
    };
    auto issue = [&](int st, const int (&ent)[4]) {
        const uint32_t dst = lds0 + (uint32_t)(st % R) * STAGE_B + (uint32_t)wave * 8192u;
        auto block_of = [&](int tok) {
            const int e = tok >> bs_shift;
            return (int64_t)(e == 0 ? ent[0] : (e == 1 ? ent[1] : (e == 2 ? ent[2] : ent[3])));
        };
        if (wave < 2) {
            int tok = ktok;
            if (64 * st + tok >= ctx) tok = 0;
            const int64_t blk = block_of(tok);
            const int off = tok & (bs - 1);
            const uint8_t* src = kc8 + ((blk * p.Hkv + hk) * 16 + 8 * wave) * (int64_t)bs * 16 + (int64_t)off * 16;
#pragma unroll
            for (int q = 0; q < 8; ++q) pa_dma16(src + (int64_t)q * bs * 16, dst + (uint32_t)q * 1024u);
        } else {
            const uint8_t* srcp[2];
#pragma unroll
            for (int par = 0; par < 2; ++par) {
                int tok = pal_dma_token(8 * wave + par, lane);
                if (64 * st + tok >= ctx) tok = 0;
                const int64_t blk = block_of(tok);
                const int off = tok & (bs - 1);
                srcp[par] = vc8 + (((blk * p.Hkv + hk) * D + pal_dma_row(8 * wave + par, lane)) * (int64_t)bs + off) * 2;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) pa_dma16(srcp[q & 1] + (int64_t)(8 * (q & ~1)) * bs * 2, dst + (uint32_t)q * 1024u);
        }
    };
    int ent[4];
#pragma unroll
    for (int st = 0; st < R - 1; ++st)
        if (st < ns) { load_ent(st, ent); issue(st, ent); }
    if (R - 1 < ns) load_ent(R - 1, ent);
    float m_run[QT], l_run[QT];
    f32x4_t o[QT][NDT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        m_run[qt] = -INFINITY; l_run[qt] = 0.f;
#pragma unroll
        for (int nt = 0; nt < NDT; ++nt) o[qt][nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    for (int i = 0; i < ns; ++i) {
        {
            const int ahead = min(ns, i + R - 1) - (i + 1);
            if (ahead >= 3) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
            else if (ahead == 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else if (ahead == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        if (i + R - 1 < ns) {
            issue(i + R - 1, ent);
            if (i + R < ns) load_ent(i + R, ent);
        }
        if (!active) continue;                                        // uniform per wave; the barrier is at the top of the loop
        const uint8_t* Kb = pfl_smem + (size_t)(i % R) * STAGE_B;
        const int tb = 64 * i;
        const bool diag = tb + 63 > cached + q0;                      // some query of the block does not see the whole stage
        // ---- K fragments of the stage: tile (ip, it), k-step j
        uint4 ka[2][2][NC];
#pragma unroll
        for (int ip = 0; ip < 2; ++ip)
#pragma unroll
            for (int it = 0; it < 2; ++it)
#pragma unroll
                for (int j = 0; j < NC; ++j) ka[ip][it][j] = *reinterpret_cast<const uint4*>(Kb + pal_k_read_off(j, kg, ip, it, c));
        uint4 pb[QT][2], pl[QT][2];                                     // P of query tile qt, pair ip: hi and lo bf16 pieces (A operands)
        float alpha[QT];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            float x[2][2][4];
            float tmax = -INFINITY;
#pragma unroll
            for (int ip = 0; ip < 2; ++ip)
#pragma unroll
                for (int it = 0; it < 2; ++it) {
                    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int j = 0; j < NC; ++j)
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, ka[ip][it][j]), __builtin_bit_cast(bf16x8_t, qf[qt][j]), acc, 0, 0, 0);
#pragma unroll
                    for (int v = 0; v < 4; ++v) x[ip][it][v] = acc[v];
                }
            if (p.softcap > 0.f) {                                    // one uniform branch for the sixteen logits of the query tile
#pragma unroll
                for (int ip = 0; ip < 2; ++ip)
#pragma unroll
                    for (int it = 0; it < 2; ++it)
#pragma unroll
                        for (int v = 0; v < 4; ++v) x[ip][it][v] = tanhf(x[ip][it][v] * p.scale / p.softcap) * p.softcap * 1.4426950408889634f;
            } else {
#pragma unroll
                for (int ip = 0; ip < 2; ++ip)
#pragma unroll
                    for (int it = 0; it < 2; ++it)
#pragma unroll
                        for (int v = 0; v < 4; ++v) x[ip][it][v] *= p.scale_log2;
            }
            if (diag) {
#pragma unroll
                for (int ip = 0; ip < 2; ++ip)
#pragma unroll
                    for (int it = 0; it < 2; ++it)
#pragma unroll
                        for (int v = 0; v < 4; ++v)
                            x[ip][it][v] = (tb + 32 * ip + 8 * kg + 4 * it + v <= pos[qt]) ? x[ip][it][v] : -INFINITY;   // causal (j < ctx follows)
            }
#pragma unroll
            for (int ip = 0; ip < 2; ++ip)
#pragma unroll
                for (int it = 0; it < 2; ++it)
#pragma unroll
                    for (int v = 0; v < 4; ++v) tmax = fmaxf(tmax, x[ip][it][v]);
            tmax = fmaxf(tmax, __shfl_xor(tmax, 16));
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
            const float m_new = fmaxf(m_run[qt], tmax);               // finite from the first stage on: key 0 is visible to every query
            alpha[qt] = exp2f(m_run[qt] - m_new);
            m_run[qt] = m_new;
            float psum = 0.f;
#pragma unroll
            for (int ip = 0; ip < 2; ++ip) {
                float e8[8];
#pragma unroll
                for (int it = 0; it < 2; ++it)
#pragma unroll
                    for (int v = 0; v < 4; ++v) { e8[4 * it + v] = exp2f(x[ip][it][v] - m_new); psum += e8[4 * it + v]; }
                const uint4 hi = make_uint4(cvt_pk_bf16(e8[0], e8[1]), cvt_pk_bf16(e8[2], e8[3]), cvt_pk_bf16(e8[4], e8[5]), cvt_pk_bf16(e8[6], e8[7]));
                float r8[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const uint32_t wv = (e & 1) ? ((&hi.x)[e >> 1] & 0xFFFF0000u) : ((&hi.x)[e >> 1] << 16);
                    r8[e] = e8[e] - __uint_as_float(wv);
                }
                pb[qt][ip] = hi;
                pl[qt][ip] = make_uint4(cvt_pk_bf16(r8[0], r8[1]), cvt_pk_bf16(r8[2], r8[3]), cvt_pk_bf16(r8[4], r8[5]), cvt_pk_bf16(r8[6], r8[7]));
            }
            l_run[qt] = fmaf(l_run[qt], alpha[qt], psum);             // this lane's share (keys 8kg.. of both pairs); summed over kg at the end
        }
        // ---- O = O * alpha + P . V : lane (channel 16 nt + c, queries 4kg+v of tile qt); query 4kg+v's alpha sits in lane 4kg+v
        float av[QT][4];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
#pragma unroll
            for (int v = 0; v < 4; ++v) av[qt][v] = __int_as_float(__builtin_amdgcn_ds_bpermute(4 * (4 * kg + v), __float_as_int(alpha[qt])));
        uint32_t vm[2][4];                                            // keys at or beyond ctx hold arbitrary bits: cleared in the B fragment
#pragma unroll
        for (int ip = 0; ip < 2; ++ip) {
            const int tk = tb + 32 * ip + 8 * kg;
#pragma unroll
            for (int e = 0; e < 4; ++e) vm[ip][e] = (tk + 2 * e < ctx ? 0x0000FFFFu : 0u) | (tk + 2 * e + 1 < ctx ? 0xFFFF0000u : 0u);
        }
#pragma unroll
        for (int nt = 0; nt < NDT; ++nt) {
            uint4 vv[2];
#pragma unroll
            for (int ip = 0; ip < 2; ++ip) {
                vv[ip] = *reinterpret_cast<const uint4*>(Kb + pal_v_read_off(16 * nt + c, ip, kg));
                vv[ip].x &= vm[ip][0]; vv[ip].y &= vm[ip][1]; vv[ip].z &= vm[ip][2]; vv[ip].w &= vm[ip][3];
            }
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                f32x4_t on = o[qt][nt];
#pragma unroll
                for (int v = 0; v < 4; ++v) on[v] *= av[qt][v];
#pragma unroll
                for (int ip = 0; ip < 2; ++ip) {
                    on = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, pb[qt][ip]), __builtin_bit_cast(bf16x8_t, vv[ip]), on, 0, 0, 0);
                    on = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, pl[qt][ip]), __builtin_bit_cast(bf16x8_t, vv[ip]), on, 0, 0, 0);
                }
                o[qt][nt] = on;
            }
        }
    }
    if (!active) return;
    // ---- epilogue: rows (queries 4kg+v of tile qt) need the sums of lane (4kg+v)
    uint16_t* out16 = static_cast<uint16_t*>(p.out);
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        float lt = l_run[qt] + __shfl_xor(l_run[qt], 16);
        lt += __shfl_xor(lt, 32);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const float lq = __shfl(lt, 4 * kg + v);
            const int ql = q0 + 16 * qt + 4 * kg + v;
            if (ql >= qlen) continue;
            const float inv = 1.f / lq;
            uint16_t* op = out16 + ((size_t)(q_begin + ql) * p.H + h) * D + c;
#pragma unroll
            for (int nt = 0; nt < NDT; ++nt) op[16 * nt] = f32_to_bf16(o[qt][nt][v] * inv);
        }
    }
}

static int g_pf_lds = 1;                                             // mi355_set_tuning(47, 0): A/B, prefill_attn_kernel (every wave fetches its own K / V) also where prefill_attn_lds_kernel fits
void mi355_prefill_set_lds(int v) { g_pf_lds = v; }

template <int DT, int D>
static void prefill_launch_src(const PrefillParams& p, int src, dim3 grid, hipStream_t st) {
    switch (src) {
        case SRC_CONTIG: hipLaunchKernelGGL((prefill_attn_kernel<DT, D, SRC_CONTIG>), grid, dim3(256), 0, st, p); break;
        case SRC_FLASH: hipLaunchKernelGGL((prefill_attn_kernel<DT, D, SRC_FLASH>), grid, dim3(256), 0, st, p); break;
        default: hipLaunchKernelGGL((prefill_attn_kernel<DT, D, SRC_PAGED>), grid, dim3(256), 0, st, p); break;
    }
}

extern "C" int mi355_prefill_attention(void* out, const void* q, const void* k, const void* v, const void* key_cache,
                                       const void* value_cache, const uint32_t* block_tables, const uint32_t* context_lens,
                                       const uint32_t* cu_seqlens_q, int32_t num_seqs, int32_t max_seqlen_q,
                                       int32_t num_heads, int32_t num_kv_heads, int32_t head_dim, int32_t block_size,
                                       int32_t max_blocks_per_seq, float scale, float softcap, int32_t layout,
                                       int32_t dtype, int64_t stream) {
    if (num_seqs <= 0 || max_seqlen_q <= 0) return 0;
    if (dtype != MI355_DTYPE_BF16 && dtype != MI355_DTYPE_F16) return (int)hipErrorInvalidValue;
    if (num_heads % num_kv_heads || head_dim > 256) return (int)hipErrorInvalidValue;
    const bool cached = key_cache != nullptr;
    if (cached && (!block_tables || !context_lens || !value_cache)) return (int)hipErrorInvalidValue;
    if (!cached && (!k || !v)) return (int)hipErrorInvalidValue;
    PrefillParams p{};
    p.out = out; p.q = q;
    p.k = cached ? key_cache : k; p.v = cached ? value_cache : v;
    p.block_tables = cached ? block_tables : nullptr;
    p.context_lens = cached ? context_lens : nullptr;
    p.cu_q = cu_seqlens_q;
    p.H = num_heads; p.Hkv = num_kv_heads; p.block_size = block_size; p.max_blocks = max_blocks_per_seq;
    p.scale = scale; p.scale_log2 = scale * 1.4426950408889634f; p.softcap = softcap > 0.f ? softcap : 0.f;
    const int src = !cached ? SRC_CONTIG : (layout == MI355_KV_FLASH ? SRC_FLASH : SRC_PAGED);
    hipStream_t st = (hipStream_t)stream;
    const bool mfma_ok = (head_dim == 64 || head_dim == 128) && (src != SRC_PAGED || block_size % 16 == 0);
    if (g_pf_lds && src == SRC_PAGED && dtype == MI355_DTYPE_BF16 && head_dim == 128 &&
        (block_size == 16 || block_size == 32 || block_size == 64)) {
        // round 4 default (first run on hardware in round 4: its tests green, prompt step 21.8 k -> 23.7 k tok/s at T = 2048): 64 queries x the
        // heads of a GQA group per workgroup, K / V through the LDS ring
        static Mi355DevOnce attr_done;
        if (!attr_done.done()) {
            (void)hipFuncSetAttribute((const void*)prefill_attn_lds_kernel<4, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32768);
            (void)hipFuncSetAttribute((const void*)prefill_attn_lds_kernel<2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 32768);
            attr_done.set();
        }
        const int G = num_heads / num_kv_heads, hgs = num_kv_heads * ((G + 3) / 4);
        // 64 queries per workgroup when that still gives two workgroups per CU to hand out heaviest first; else 32 (ring of two stages).
        // Measured (round 4, Llama-3-8B prompt step, one box): T = 2048 (256 workgroups of 64 queries = one round, the launch lasts as long
        // as its heaviest block) 37.8 k -> 39.8 k tok/s with 32-query blocks; T = 4096 38.9 k either way -- and 34.7 k before the grid was
        // ordered heaviest block first.
        const int64_t nwg64 = (int64_t)((max_seqlen_q + 63) / 64) * hgs * num_seqs;
        if (nwg64 >= 512) {
            dim3 grid(hgs, (max_seqlen_q + 63) / 64, num_seqs);
            hipLaunchKernelGGL((prefill_attn_lds_kernel<4, 4>), grid, dim3(256), 4 * 32768, st, p, block_tables, context_lens, cu_seqlens_q);
        } else {
            dim3 grid(hgs, (max_seqlen_q + 31) / 32, num_seqs);
            hipLaunchKernelGGL((prefill_attn_lds_kernel<2, 2>), grid, dim3(256), 2 * 32768, st, p, block_tables, context_lens, cu_seqlens_q);
        }
        return (int)hipGetLastError();
    }
    if (mfma_ok) {
        dim3 grid((max_seqlen_q + 127) / 128, num_heads, num_seqs);
        if (dtype == MI355_DTYPE_BF16) {
            if (head_dim == 128) prefill_launch_src<MI355_DTYPE_BF16, 128>(p, src, grid, st);
            else prefill_launch_src<MI355_DTYPE_BF16, 64>(p, src, grid, st);
        } else {
            if (head_dim == 128) prefill_launch_src<MI355_DTYPE_F16, 128>(p, src, grid, st);
            else prefill_launch_src<MI355_DTYPE_F16, 64>(p, src, grid, st);
        }
    } else {
        if (src == SRC_PAGED && head_dim % 8) return (int)hipErrorInvalidValue;
        dim3 grid(max_seqlen_q, num_heads, num_seqs);
        if (dtype == MI355_DTYPE_BF16) hipLaunchKernelGGL((prefill_attn_generic_kernel<MI355_DTYPE_BF16>), grid, dim3(64), 0, st, p, head_dim, src);
        else hipLaunchKernelGGL((prefill_attn_generic_kernel<MI355_DTYPE_F16>), grid, dim3(64), 0, st, p, head_dim, src);
    }
    return (int)hipGetLastError();
}

/* `sliding_window` of PagedAttention::new on prompt steps (attention.rs:566-575,888-897; layers/mask.rs:22-27): the same operator with the
 * causal mask narrowed to the last `sliding_window` keys of every query.  No BASELINE model carries a window, so this is the correctness
 * path: the one-wave-per-(query, head) kernel, every head size / layout / 16-bit dtype.  sliding_window <= 0 = mi355_prefill_attention. */
extern "C" int mi355_prefill_attention_window(void* out, const void* q, const void* k, const void* v, const void* key_cache,
                                              const void* value_cache, const uint32_t* block_tables, const uint32_t* context_lens,
                                              const uint32_t* cu_seqlens_q, int32_t num_seqs, int32_t max_seqlen_q,
                                              int32_t num_heads, int32_t num_kv_heads, int32_t head_dim, int32_t block_size,
                                              int32_t max_blocks_per_seq, float scale, float softcap, int32_t layout,
                                              int32_t dtype, int32_t sliding_window, int64_t stream) {
    if (sliding_window <= 0)
        return mi355_prefill_attention(out, q, k, v, key_cache, value_cache, block_tables, context_lens, cu_seqlens_q, num_seqs, max_seqlen_q,
                                       num_heads, num_kv_heads, head_dim, block_size, max_blocks_per_seq, scale, softcap, layout, dtype, stream);
    if (num_seqs <= 0 || max_seqlen_q <= 0) return 0;
    if (dtype != MI355_DTYPE_BF16 && dtype != MI355_DTYPE_F16) return (int)hipErrorInvalidValue;
    if (num_kv_heads <= 0 || num_heads % num_kv_heads || head_dim > 256 || head_dim <= 0) return (int)hipErrorInvalidValue;
    const bool cached = key_cache != nullptr;
    if (cached && (!block_tables || !context_lens || !value_cache)) return (int)hipErrorInvalidValue;
    if (!cached && (!k || !v)) return (int)hipErrorInvalidValue;
    if (!out || !q || !cu_seqlens_q) return (int)hipErrorInvalidValue;
    PrefillParams p{};
    p.out = out; p.q = q;
    p.k = cached ? key_cache : k; p.v = cached ? value_cache : v;
    p.block_tables = cached ? block_tables : nullptr;
    p.context_lens = cached ? context_lens : nullptr;
    p.cu_q = cu_seqlens_q;
    p.H = num_heads; p.Hkv = num_kv_heads; p.block_size = block_size; p.max_blocks = max_blocks_per_seq;
    p.scale = scale; p.scale_log2 = scale * 1.4426950408889634f; p.softcap = softcap > 0.f ? softcap : 0.f;
    p.window = sliding_window;
    const int src = !cached ? SRC_CONTIG : (layout == MI355_KV_FLASH ? SRC_FLASH : SRC_PAGED);
    if (src == SRC_PAGED && head_dim % 8) return (int)hipErrorInvalidValue;
    dim3 grid(max_seqlen_q, num_heads, num_seqs);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MI355_DTYPE_BF16) hipLaunchKernelGGL((prefill_attn_generic_kernel<MI355_DTYPE_BF16>), grid, dim3(64), 0, st, p, head_dim, src);
    else hipLaunchKernelGGL((prefill_attn_generic_kernel<MI355_DTYPE_F16>), grid, dim3(64), 0, st, p, head_dim, src);
    return (int)hipGetLastError();
}

// prefill over an fp8 (e4m3fn) KV cache (PAGED layout, K x = 16): every key -- cached prefix and the current chunk,
// already written by mi355_reshape_and_cache_fp8 -- is read from the cache.  MFMA flash kernel for head sizes 64 / 128 (round 3;
// attention.rs:574,896), the generic kernel otherwise (and as the A/B: tuning key 43).
static int g_pf_fp8_generic = 0;
void mi355_prefill_set_fp8_generic(int v) { g_pf_fp8_generic = v; }

// Round 6: the e4m3 cache blocks of the sequences of a prompt step as bf16 (exact: 3 mantissa bits), in the bf16 cache layouts
// K [NBt, Hkv, D/8, bs, 8] / V [NBt, Hkv, D, bs], temp block seq * max_blocks + i, with the table that says so -- so that the step's attention
// runs through prefill_attn_lds_kernel (K / V through the LDS ring: 290 TFLOP/s on the Llama prompt step) instead of the register-fed
// prefill_attn_kernel<.., SRC_PAGED8> (110 TFLOP/s, 51 % of a 16 k-token Mixtral prompt: profiles/r06_moe_prompt_device_ab.txt).  One pass
// over ctx x Hkv x D x 2 bytes per layer (33 MB read, 67 MB written at 16 k tokens: ~25 us next to 10 ms of attention).
__global__ void __launch_bounds__(256) pf_fp8_blocks_to_bf16_kernel(uint16_t* __restrict__ tk, uint16_t* __restrict__ tv, uint32_t* __restrict__ tbt,
                                                                    const uint8_t* __restrict__ kc, const uint8_t* __restrict__ vc,
                                                                    const uint32_t* __restrict__ bt, const uint32_t* __restrict__ ctx,
                                                                    int Hkv, int D, int bs, int max_blocks) {
    const int i = blockIdx.x, hk = blockIdx.y, seq = blockIdx.z;
    if (hk == 0 && threadIdx.x == 0) tbt[(size_t)seq * max_blocks + i] = (uint32_t)(seq * max_blocks + i);
    // tokens past the context become ZERO, up to the end of the 128-token span the last stage of the ring may touch (a cache block holds
    // finite stale values there; fresh scratch memory and e4m3 bytes need not: 0 x NaN in P.V would poison a row); later blocks: untouched
    const int n_ctx = (int)ctx[seq], live = n_ctx - i * bs;            // tokens of this block inside the context (<= 0: none)
    if ((int64_t)i * bs >= (((int64_t)n_ctx + 127) / 128) * 128 + 128) return;
    const size_t src = live > 0 ? ((size_t)bt[(size_t)seq * max_blocks + i] * Hkv + hk) * D * bs : 0;      // bytes (the table entry of a block past the context may be anything)
    const size_t dst = ((size_t)(seq * max_blocks + i) * Hkv + hk) * D * bs;                // elements
    for (int u = threadIdx.x; u < (D / 8) * bs; u += 256) {            // K: (d8, t) -> 8 consecutive d of token t
        const int d8 = u / bs, t = u - d8 * bs;
        uint4 o = make_uint4(0, 0, 0, 0);
        if (t < live) {
            const uint2 raw = *reinterpret_cast<const uint2*>(kc + src + ((size_t)(d8 >> 1) * bs + t) * 16 + (d8 & 1) * 8);
            const uint2 lo = pf_fp8x4_to_bf16x4(raw.x), hi = pf_fp8x4_to_bf16x4(raw.y);
            o = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
        *reinterpret_cast<uint4*>(tk + dst + ((size_t)d8 * bs + t) * 8) = o;
    }
    for (int u = threadIdx.x; u < D * bs / 8; u += 256) {              // V: [D][bs] in both caches; 8 consecutive tokens of one channel
        const int t0 = (u * 8) % bs;
        uint32_t w[4] = {0, 0, 0, 0};
        if (t0 < live) {
            const uint2 raw = *reinterpret_cast<const uint2*>(vc + src + (size_t)u * 8);
            const uint2 lo = pf_fp8x4_to_bf16x4(raw.x), hi = pf_fp8x4_to_bf16x4(raw.y);
            w[0] = lo.x; w[1] = lo.y; w[2] = hi.x; w[3] = hi.y;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (t0 + 2 * e >= live) w[e] = 0;
                else if (t0 + 2 * e + 1 >= live) w[e] &= 0xFFFFu;
            }
        }
        *reinterpret_cast<uint4*>(tv + dst + (size_t)u * 8) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}
// MI355_PF_FP8_VIA_BF16=0 (A/B runs): the register-fed e4m3 kernel as in rounds 3-5
static const int g_pf_fp8_via_bf16_env = []() { const char* e = getenv("MI355_PF_FP8_VIA_BF16"); return e && e[0] == '0' ? 0 : 1; }();
extern "C" int mi355_prefill_attention_fp8(void* out, const void* q, const void* key_cache, const void* value_cache,
                                           const uint32_t* block_tables, const uint32_t* context_lens,
                                           const uint32_t* cu_seqlens_q, int32_t num_seqs, int32_t max_seqlen_q,
                                           int32_t num_heads, int32_t num_kv_heads, int32_t head_dim, int32_t block_size,
                                           int32_t max_blocks_per_seq, float scale, float softcap, float k_scale,
                                           float v_scale, int32_t dtype, int64_t stream) {
    if (num_seqs <= 0 || max_seqlen_q <= 0) return 0;
    if (dtype != MI355_DTYPE_BF16 || (head_dim % 16) || head_dim > 256 || num_heads % num_kv_heads) return (int)hipErrorInvalidValue;
    if (!key_cache || !value_cache || !block_tables || !context_lens) return (int)hipErrorInvalidValue;
    PrefillParams p{};
    p.out = out; p.q = q; p.k = key_cache; p.v = value_cache;
    p.block_tables = block_tables; p.context_lens = context_lens; p.cu_q = cu_seqlens_q;
    p.H = num_heads; p.Hkv = num_kv_heads; p.block_size = block_size; p.max_blocks = max_blocks_per_seq;
    p.scale = scale; p.scale_log2 = scale * 1.4426950408889634f; p.softcap = softcap > 0.f ? softcap : 0.f;
    p.kv8 = 1; p.k_scale = k_scale; p.v_scale = v_scale;
    {
        // through the LDS-fed bf16 kernel (see pf_fp8_blocks_to_bf16_kernel): head size 128, a unit value scale (k_scale folds into the
        // logit scale as below), and a temporary of at most 2 GiB
        const size_t elems = (size_t)num_seqs * max_blocks_per_seq * num_kv_heads * head_dim * block_size;
        if (g_pf_fp8_via_bf16_env && g_pf_lds && !g_pf_fp8_generic && head_dim == 128 && v_scale == 1.0f &&
            (block_size == 16 || block_size == 32 || block_size == 64) && elems * 4 <= ((size_t)2 << 30)) {
            void* ws = nullptr;
            const size_t tb_bytes = ((size_t)num_seqs * max_blocks_per_seq * 4 + 255) / 256 * 256;
            const int rc = mi355_scratch_get(&ws, MI355_SCR_PF_FP8, elems * 4 + tb_bytes, (hipStream_t)stream, false);
            if (rc == 0) {
                uint16_t* tk = static_cast<uint16_t*>(ws);
                uint16_t* tv = tk + elems;
                uint32_t* tbt = reinterpret_cast<uint32_t*>(tv + elems);
                hipLaunchKernelGGL(pf_fp8_blocks_to_bf16_kernel, dim3(max_blocks_per_seq, num_kv_heads, num_seqs), dim3(256), 0, (hipStream_t)stream, tk, tv, tbt,
                                   static_cast<const uint8_t*>(key_cache), static_cast<const uint8_t*>(value_cache), block_tables, context_lens,
                                   num_kv_heads, head_dim, block_size, max_blocks_per_seq);
                return mi355_prefill_attention(out, q, nullptr, nullptr, tk, tv, tbt, context_lens, cu_seqlens_q, num_seqs, max_seqlen_q, num_heads,
                                               num_kv_heads, head_dim, block_size, max_blocks_per_seq, scale * k_scale, softcap, MI355_KV_PAGED,
                                               MI355_DTYPE_BF16, stream);
            }
        }
    }
    if ((head_dim == 64 || head_dim == 128) && block_size % 16 == 0 && !g_pf_fp8_generic) {
        // the MFMA flash kernel over the e4m3 cache: bytes converted to bf16 fragments on the way in (exact), k_scale folded into the
        // logit scale, v_scale into the normalisation
        PrefillParams q8 = p;
        q8.scale = scale * k_scale; q8.scale_log2 = scale * k_scale * 1.4426950408889634f;
        dim3 grid8((max_seqlen_q + 127) / 128, num_heads, num_seqs);
        if (head_dim == 128) hipLaunchKernelGGL((prefill_attn_kernel<MI355_DTYPE_BF16, 128, SRC_PAGED8>), grid8, dim3(256), 0, (hipStream_t)stream, q8);
        else hipLaunchKernelGGL((prefill_attn_kernel<MI355_DTYPE_BF16, 64, SRC_PAGED8>), grid8, dim3(256), 0, (hipStream_t)stream, q8);
        return (int)hipGetLastError();
    }
    dim3 grid(max_seqlen_q, num_heads, num_seqs);
    hipLaunchKernelGGL((prefill_attn_generic_kernel<MI355_DTYPE_BF16>), grid, dim3(64), 0, (hipStream_t)stream, p, head_dim, SRC_PAGED);
    return (int)hipGetLastError();
}
