// Paged-attention decode for gfx950 (MI355X): K2 `paged_attention_v1`, K3 `paged_attention_v2` + reduce.
//
// Reference boundary: attention_rs::PagedAttention::forward as called from
//   src/openai/models/layers/attention.rs:707-719,983-995 with the InputMetadata of
//   src/openai/pipelines/inputs.rs:552-568 (block_tables u32 [B,max_blocks], context_lens u32 [B]).
// Math (oracle): NaiveAttention, src/openai/models/mod.rs:1288-1306 -- softmax(q.k^T*scale [softcap]).v
// with GQA expansion (:1240-1247); fp32 scores / softmax / accumulation, 16-bit output.
//
// Design (HBM-bound: every live K/V byte is read exactly once):
//   * one workgroup (4 waves) per (kv head, sequence, context partition); the G = H/Hkv query heads of
//     the GQA group share each K/V byte out of registers,
//   * a token's head row (D 16-bit elements) is spread over LPT = D/8 lanes, 16 B per lane -> a wave
//     reads 64/LPT tokens per instruction as fully used 128-B lines (FLASH layout) ,
//   * q.k partial dots are reduced over the LPT lanes with DPP row operations (no LDS),
//   * chunked online softmax (UNR tokens per chunk, one rescale per chunk),
//   * lane-group states merge with ds_bpermute shuffles, wave states through 8.3 KB of LDS,
//   * v2: partitions write (normalised out, max_logit, exp_sum); a tiny reduce kernel merges them.
#include "common.h"
#include "pa_lds_layout.h"
#include "scratch.h"
#include <type_traits>
#include "../../include/mi355_vllm.h"


template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
// all-reduce sum over aligned groups of LPT lanes
template <int LPT>
__device__ __forceinline__ float lpt_sum(float v) {
    if constexpr (LPT >= 32) v += __shfl_xor(v, 16, 64);
    if constexpr (LPT >= 16) v += dpp_mov<0x140>(v);   // row_mirror      : lane i <-> 15-i
    if constexpr (LPT >= 8) v += dpp_mov<0x141>(v);    // row_half_mirror : lane i <-> 7-i
    v += dpp_mov<0x4E>(v);                             // quad_perm [2,3,0,1]
    v += dpp_mov<0xB1>(v);                             // quad_perm [1,0,3,2]
    return v;
}

template <int KVT>
__device__ __forceinline__ void unpack8(const uint4& w, float (&f)[8]) {
    const uint32_t u[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if constexpr (KVT == MI355_DTYPE_BF16) {
            f[2 * i] = bf16lo_to_f32(u[i]);
            f[2 * i + 1] = bf16hi_to_f32(u[i]);
        } else {
            f[2 * i] = f16_bits_to_f32((uint16_t)(u[i] & 0xFFFF));
            f[2 * i + 1] = f16_bits_to_f32((uint16_t)(u[i] >> 16));
        }
    }
}

struct PAParams {
    void* out;                 // v1: [B,H,D] 16-bit ; v2: unused
    float* tmp_out;            // v2: [B,H,P,D] f32 (normalised per partition)
    float* max_logits;         // v2: [B,H,P]
    float* exp_sums;           // v2: [B,H,P]
    const void* q;             // [B,H,D] 16-bit
    const void* kc;            // FLASH: [NB, bs, Hkv, D]
    const void* vc;
    const uint32_t* block_tables;   // [B, max_blocks]
    const uint32_t* context_lens;   // [B]
    int H, Hkv, D, block_size, max_blocks;
    int partition_size;        // tokens per partition (v1: >= max context)
    int max_partitions;        // P (stride of the v2 temporaries); 1 for v1
    float scale, softcap;      // softcap <= 0 -> disabled
    int64_t q_stride;          // elements between sequences in q (H*D when contiguous)
    unsigned* arrive;          // [B*Hkv] arrival counters (zero between launches) or null: fused partition merge
    int kv8;                   // 1: the cache holds OCP e4m3fn bytes (K layout x = 16), value = byte * {k,v}_scale
    float k_scale, v_scale;
};

// 4 e4m3fn bytes -> 4 bf16 (exact: 3 mantissa bits), 8 bytes -> 8 bf16
__device__ __forceinline__ uint2 fp8x4_to_bf16x4(uint32_t w) {
    typedef float pa_f32x2 __attribute__((ext_vector_type(2)));
    const pa_f32x2 a = __builtin_amdgcn_cvt_pk_f32_fp8((int)w, false), b = __builtin_amdgcn_cvt_pk_f32_fp8((int)w, true);
    return make_uint2(cvt_pk_bf16(a.x, a.y), cvt_pk_bf16(b.x, b.y));
}
__device__ __forceinline__ float fp8_to_f32(uint8_t b) {
    typedef float pa_f32x2 __attribute__((ext_vector_type(2)));
    const pa_f32x2 a = __builtin_amdgcn_cvt_pk_f32_fp8((int)b, false);
    return a.x;
}

// NWV waves per workgroup.  The partitioned (v2) path runs ONE wave per workgroup (no LDS, no barrier: a
// partition is what a single wave streams with UNR loads in flight); v1 keeps 4 waves merged through LDS.
template <int LPT, int GP, int KVT, bool PARTITIONED, int NWV, int UNR>
__global__ void __launch_bounds__(64 * NWV) paged_attn_flash_kernel(const PAParams p) {
    constexpr int TW = 64 / LPT;            // token groups per wave
    constexpr int TI = NWV * TW;            // tokens per workgroup iteration
    const int hk = blockIdx.x, b = blockIdx.y, part = blockIdx.z;
    const int ctx = (int)p.context_lens[b];
    const int t0 = part * p.partition_size;
    if (t0 >= ctx) return;
    const int t1 = min(ctx, t0 + p.partition_size);
    const int G = p.H / p.Hkv;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane % LPT, tg = wave * TW + lane / LPT;
    const bool dact = sub * 8 < p.D;        // head_dim not a power of two (e.g. 80): upper lanes idle
    const uint16_t* kc = static_cast<const uint16_t*>(p.kc);
    const uint16_t* vc = static_cast<const uint16_t*>(p.vc);
    const uint32_t* bt = p.block_tables + (int64_t)b * p.max_blocks;

    // q (pre-scaled) for the G heads of this kv head, this lane's 8 channels
    float q[GP][8];
#pragma unroll
    for (int g = 0; g < GP; ++g) {
        if (g < G && dact) {
            const uint16_t* qp = static_cast<const uint16_t*>(p.q) + (int64_t)b * p.q_stride +
                                 (int64_t)(hk * G + g) * p.D + sub * 8;
            float f[8];
            unpack8<KVT>(*reinterpret_cast<const uint4*>(qp), f);
#pragma unroll
            for (int e = 0; e < 8; ++e) q[g][e] = f[e] * p.scale;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) q[g][e] = 0.f;
        }
    }

    float m[GP], l[GP], acc[GP][8];
#pragma unroll
    for (int g = 0; g < GP; ++g) {
        m[g] = -1e30f;
        l[g] = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[g][e] = 0.f;
    }

    const int64_t tok_stride = (int64_t)p.Hkv * p.D;            // elements between tokens of a block
    for (int base = t0 + tg; base < t1; base += TI * UNR) {
        uint4 kw[UNR], vw[UNR];
        bool valid[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int tok = base + u * TI;
            valid[u] = tok < t1;
            kw[u] = make_uint4(0, 0, 0, 0);
            vw[u] = make_uint4(0, 0, 0, 0);
            if (valid[u] && dact) {
                const int64_t blk = (int64_t)bt[tok / p.block_size];
                const int64_t off = (blk * p.block_size + (tok % p.block_size)) * tok_stride +
                                    (int64_t)hk * p.D + sub * 8;
                kw[u] = *reinterpret_cast<const uint4*>(kc + off);
                vw[u] = *reinterpret_cast<const uint4*>(vc + off);
            }
        }
        float s[UNR][GP];
        float mx[GP];
#pragma unroll
        for (int g = 0; g < GP; ++g) mx[g] = m[g];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            float kf[8];
            unpack8<KVT>(kw[u], kf);
#pragma unroll
            for (int g = 0; g < GP; ++g) {
                float d = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) d = fmaf(q[g][e], kf[e], d);
                d = lpt_sum<LPT>(d);
                if (p.softcap > 0.f) d = tanhf(d / p.softcap) * p.softcap;
                s[u][g] = d;
                if (valid[u]) mx[g] = fmaxf(mx[g], d);
            }
        }
#pragma unroll
        for (int g = 0; g < GP; ++g) {
            const float alpha = __expf(m[g] - mx[g]);
            m[g] = mx[g];
            l[g] *= alpha;
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[g][e] *= alpha;
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            float vf[8];
            unpack8<KVT>(vw[u], vf);
#pragma unroll
            for (int g = 0; g < GP; ++g) {
                const float pr = valid[u] ? __expf(s[u][g] - m[g]) : 0.f;
                l[g] += pr;
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[g][e] = fmaf(pr, vf[e], acc[g][e]);
            }
        }
    }

    // ---- merge the TW lane groups of this wave (same `sub`, different tokens)
#pragma unroll
    for (int g = 0; g < GP; ++g) {
        float M = m[g];
#pragma unroll
        for (int o = LPT; o < 64; o <<= 1) M = fmaxf(M, __shfl_xor(M, o, 64));
        const float w = __expf(m[g] - M);
        float ll = l[g] * w;
#pragma unroll
        for (int o = LPT; o < 64; o <<= 1) ll += __shfl_xor(ll, o, 64);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float a = acc[g][e] * w;
#pragma unroll
            for (int o = LPT; o < 64; o <<= 1) a += __shfl_xor(a, o, 64);
            acc[g][e] = a;
        }
        m[g] = M;
        l[g] = ll;
    }

    if constexpr (NWV == 1) {
        // single wave: lanes 0..LPT-1 hold the merged state of their 8 channels -> write the partial directly
        if (lane < LPT && dact) {
#pragma unroll
            for (int g = 0; g < GP; ++g) {
                if (g >= G) break;
                const int h = hk * G + g;
                const float inv = l[g] > 0.f ? 1.f / l[g] : 0.f;
                if constexpr (PARTITIONED) {
                    const int64_t pi = ((int64_t)b * p.H + h) * p.max_partitions + part;
                    float4* tp = reinterpret_cast<float4*>(p.tmp_out + pi * p.D + sub * 8);
                    tp[0] = make_float4(acc[g][0] * inv, acc[g][1] * inv, acc[g][2] * inv, acc[g][3] * inv);
                    tp[1] = make_float4(acc[g][4] * inv, acc[g][5] * inv, acc[g][6] * inv, acc[g][7] * inv);
                    if (sub == 0) { p.max_logits[pi] = m[g]; p.exp_sums[pi] = l[g]; }
                } else {
                    uint16_t* op = static_cast<uint16_t*>(p.out) + ((int64_t)b * p.H + h) * p.D + sub * 8;
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        op[e] = (KVT == MI355_DTYPE_BF16) ? f32_to_bf16(acc[g][e] * inv) : f32_to_f16_bits(acc[g][e] * inv);
                }
            }
        }
        return;
    }
    // ---- merge the NWV waves through LDS
    constexpr int DP = LPT * 8;                        // padded head dim
    __shared__ float s_acc[4][GP][DP];
    __shared__ float s_m[4][GP], s_l[4][GP];
    if (lane < LPT) {
#pragma unroll
        for (int g = 0; g < GP; ++g) {
#pragma unroll
            for (int e = 0; e < 8; ++e) s_acc[wave][g][sub * 8 + e] = acc[g][e];
            if (lane == 0) { s_m[wave][g] = m[g]; s_l[wave][g] = l[g]; }
        }
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < G * p.D; idx += 64 * NWV) {
        const int g = idx / p.D, d = idx % p.D;
        float M = fmaxf(fmaxf(s_m[0][g], s_m[1][g]), fmaxf(s_m[2][g], s_m[3][g]));
        float num = 0.f, den = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float f = __expf(s_m[w][g] - M);
            num += s_acc[w][g][d] * f;
            den += s_l[w][g] * f;
        }
        const float o = num / den;
        const int h = hk * G + g;
        if constexpr (PARTITIONED) {
            const int64_t pi = ((int64_t)b * p.H + h) * p.max_partitions + part;
            p.tmp_out[pi * p.D + d] = o;
            if (d == 0) { p.max_logits[pi] = M; p.exp_sums[pi] = den; }
        } else {
            uint16_t* op = static_cast<uint16_t*>(p.out) + ((int64_t)b * p.H + h) * p.D + d;
            *op = (KVT == MI355_DTYPE_BF16) ? f32_to_bf16(o) : f32_to_f16_bits(o);
        }
    }
}

// v2 reduce: grid (H, B); merges the partitions of one (sequence, head).  Partition statistics are pulled
// into LDS in parallel (no serial dependent-load chain), then every thread owns output channels and sums the
// partitions with independent loads.
template <int KVT>
__global__ void __launch_bounds__(512) paged_attn_reduce_kernel(void* __restrict__ out, const float* __restrict__ tmp_out,
                                                                const float* __restrict__ max_logits,
                                                                const float* __restrict__ exp_sums,
                                                                const uint32_t* __restrict__ context_lens, int H,
                                                                int D, int partition_size, int max_partitions) {
    extern __shared__ float s_w[];                          // [P] merge weights, then [NG][DP] partial outputs
    __shared__ float red[16];
    const int h = blockIdx.x, b = blockIdx.y;
    const int64_t base = ((int64_t)b * H + h) * max_partitions;
    // statistics of every possible partition are fetched while context_lens is still in flight
    float ml0 = (threadIdx.x < max_partitions) ? max_logits[base + threadIdx.x] : -1e30f;
    const int ctx = (int)context_lens[b];
    const int P = min((ctx + partition_size - 1) / partition_size, max_partitions);   // never beyond the launched partitions
    float M = -1e30f;
    for (int i = threadIdx.x; i < P; i += blockDim.x) {
        const float ml = (i == (int)threadIdx.x) ? ml0 : max_logits[base + i];
        s_w[i] = ml;
        M = fmaxf(M, ml);
    }
    M = wave_max(M);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = M;
    __syncthreads();
    M = red[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) M = fmaxf(M, red[w]);
    __syncthreads();
    float den = 0.f;
    for (int i = threadIdx.x; i < P; i += blockDim.x) {
        const float w = exp_sums[base + i] * __expf(s_w[i] - M);
        s_w[i] = w;
        den += w;
    }
    den = block_sum(den, red);                              // barrier inside publishes s_w
    const float inv = (P > 0 && den > 0.f) ? 1.f / den : 0.f;
    // the partitions are dealt round-robin to NG thread groups (one output channel per thread, 8 independent loads
    // in flight each): the partial rows come from other XCDs' writes, i.e. from memory -- latency, not bandwidth
    const int DP = D <= 64 ? 64 : (D <= 128 ? 128 : 256);
    const int NG = blockDim.x / DP, g = threadIdx.x / DP, d = threadIdx.x % DP;
    float acc = 0.f;
    if (d < D) {
        const float* tp = tmp_out + base * D + d;
        float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int i = g;
        for (; i + 7 * NG < P; i += 8 * NG) {
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] = fmaf(tp[(int64_t)(i + u * NG) * D], s_w[i + u * NG], a[u]);
        }
        for (; i < P; i += NG) a[0] = fmaf(tp[(int64_t)i * D], s_w[i], a[0]);
        acc = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    }
    __syncthreads();                                        // everybody is done reading the weights
    float* s_o = s_w;                                       // reuse: [NG][DP]
    s_o[g * DP + d] = acc;
    __syncthreads();
    if (g == 0 && d < D) {
        float o = 0.f;
        for (int k = 0; k < NG; ++k) o += s_o[k * DP + d];
        o *= inv;
        uint16_t* op = static_cast<uint16_t*>(out) + ((int64_t)b * H + h) * D + d;
        *op = (KVT == MI355_DTYPE_BF16) ? f32_to_bf16(o) : f32_to_f16_bits(o);
    }
}

// ------------------------------------------------------------------------------------------------
// PAGED (vLLM) layout: K [NB,Hkv,D/8,bs,8], V [NB,Hkv,D,bs] (16-bit).  Functional kernel: one workgroup
// per (head, sequence); lane = token, logits staged in LDS.  Correct for every block size / head dim;
// the FLASH layout above is the tuned path (cache_engine.rs:188-193 makes FLASH the cuda-build default).
template <int KVT>
__global__ void __launch_bounds__(256) paged_attn_paged_layout_kernel(const PAParams p) {
    extern __shared__ float s_logits[];                  // [partition tokens]
    __shared__ float red[16];
    __shared__ float s_q[256];
    const int h = blockIdx.x, b = blockIdx.y, part = blockIdx.z;
    const int ctx = (int)p.context_lens[b];
    const int t0 = part * p.partition_size;
    if (t0 >= ctx) return;
    const int t1 = min(ctx, t0 + p.partition_size);
    const int n = t1 - t0;
    const int G = p.H / p.Hkv, hk = h / G, D = p.D, bs = p.block_size;
    const uint16_t* kc = static_cast<const uint16_t*>(p.kc);
    const uint16_t* vc = static_cast<const uint16_t*>(p.vc);
    const uint32_t* bt = p.block_tables + (int64_t)b * p.max_blocks;
    const uint16_t* qp = static_cast<const uint16_t*>(p.q) + (int64_t)b * p.q_stride + (int64_t)h * D;
    for (int d = threadIdx.x; d < D; d += blockDim.x)
        s_q[d] = ((KVT == MI355_DTYPE_BF16) ? bf16_to_f32(qp[d]) : f16_bits_to_f32(qp[d])) * p.scale;
    __syncthreads();
    float lmax = -1e30f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int tok = t0 + i;
        const int64_t blk = (int64_t)bt[tok / bs];
        const int off = tok % bs;
        float s = 0.f;
        if (p.kv8) {                                       // K [NB,Hkv,D/16,bs,16] e4m3fn
            const uint8_t* kb8 = static_cast<const uint8_t*>(p.kc) + ((blk * p.Hkv + hk) * (D / 16)) * (int64_t)bs * 16 + (int64_t)off * 16;
            for (int dg = 0; dg < D / 16; ++dg) {
                const uint4 w = *reinterpret_cast<const uint4*>(kb8 + (int64_t)dg * bs * 16);
                const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                for (int e = 0; e < 16; ++e) s = fmaf(s_q[dg * 16 + e], fp8_to_f32((uint8_t)(ww[e >> 2] >> (8 * (e & 3)))) * p.k_scale, s);
            }
        } else {
            const uint16_t* kb = kc + ((blk * p.Hkv + hk) * (D / 8)) * (int64_t)bs * 8 + (int64_t)off * 8;
            for (int dg = 0; dg < D / 8; ++dg) {
                float f[8];
                unpack8<KVT>(*reinterpret_cast<const uint4*>(kb + (int64_t)dg * bs * 8), f);
#pragma unroll
                for (int e = 0; e < 8; ++e) s = fmaf(s_q[dg * 8 + e], f[e], s);
            }
        }
        if (p.softcap > 0.f) s = tanhf(s / p.softcap) * p.softcap;
        s_logits[i] = s;
        lmax = fmaxf(lmax, s);
    }
    {   // block max
        const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
        lmax = wave_max(lmax);
        if (lane == 0) red[wid] = lmax;
        __syncthreads();
        lmax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        __syncthreads();
    }
    float lsum = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float e = __expf(s_logits[i] - lmax);
        s_logits[i] = e;
        lsum += e;
    }
    lsum = block_sum(lsum, red);                         // includes the barrier that publishes s_logits
    for (int d = threadIdx.x; d < D; d += blockDim.x) {
        float a = 0.f;
        for (int i = 0; i < n; ++i) {
            const int tok = t0 + i;
            const int64_t blk = (int64_t)bt[tok / bs];
            const int64_t vo = ((blk * p.Hkv + hk) * D + d) * (int64_t)bs + tok % bs;
            float vf;
            if (p.kv8) vf = fp8_to_f32(static_cast<const uint8_t*>(p.vc)[vo]) * p.v_scale;
            else { const uint16_t vv = vc[vo]; vf = (KVT == MI355_DTYPE_BF16) ? bf16_to_f32(vv) : f16_bits_to_f32(vv); }
            a = fmaf(s_logits[i], vf, a);
        }
        const float o = a / lsum;
        if (p.max_partitions > 1) {
            const int64_t pi = ((int64_t)b * p.H + h) * p.max_partitions + part;
            p.tmp_out[pi * D + d] = o;
            if (d == 0) { p.max_logits[pi] = lmax; p.exp_sums[pi] = lsum; }
        } else {
            uint16_t* op = static_cast<uint16_t*>(p.out) + ((int64_t)b * p.H + h) * D + d;
            *op = (KVT == MI355_DTYPE_BF16) ? f32_to_bf16(o) : f32_to_f16_bits(o);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// MFMA decode attention on the PAGED (vLLM) layout -- the layout candle-vllm builds use when no flash-attn
// feature is enabled (cache_engine.rs:188-193), i.e. what a ROCm build of the reference would allocate.
// K [NB,Hkv,D/8,bs,8] hands the QK^T A-fragment (16 B = 8 channels of one token) and V [NB,Hkv,D,bs] the
// P.V B-fragment (8 B = 4 tokens of one channel) straight from global memory: no LDS, no transpose.
//   * one wave per (kv head, sequence, partition of 16*NT tokens); the G <= 16 query heads of the GQA group are
//     the 16 MFMA columns (swapped product S^T = K.Q^T, so the softmax output is already the A operand of P.V);
//   * the whole partition's scores stay in registers -> one max per partition, no online rescale;
//   * P is rounded to bf16 for the P.V MFMA (fp32 accumulate), output normalised per partition;
//   * all K and V loads of the partition are issued before the first MFMA.
// one partition [t0, t1) of (sequence b, kv head hk) on one wave: unnormalised O (lane: channel 16nt + (lane&15), heads
// 4(lane>>4)+v), the partition's row maxima m and bf16-rounded probability sums lsum (valid in lane == head)
template <int D32, int NT, bool KV8>
__device__ __forceinline__ void pa_mfma_partition(const PAParams& p, const int b, const int hk, const int t0, const int t1,
                                                  const int lane, f32x4_t (&o)[2 * D32], float& m, float& lsum, const int blk_s = -1) {
    constexpr int D = 32 * D32, NTD = D / 16;
    const int G = p.H / p.Hkv, bs = p.block_size;
    const int c = lane & 15, kg = lane >> 4;
    const uint16_t* kc = static_cast<const uint16_t*>(p.kc);
    const uint16_t* vc = static_cast<const uint16_t*>(p.vc);

    // Token order inside a 32-token pair of tiles is chosen for the MEMORY side: tile A row 4g+v <-> token 8g+v, tile B
    // row 4g+v <-> token 8g+4+v.  Then (i) a lane's probabilities of tiles A and B are the 8 consecutive tokens
    // 8kg..8kg+7 -> the A fragment of ONE 16x16x32 P.V MFMA, register for register, and (ii) V can be fetched as
    // 16 contiguous bytes per lane with the four lanes of a quad on one channel row (quad = 64 contiguous bytes)
    // and moved to the MFMA's (channel = lane&15, token group = lane>>4) placement by ds_bpermute.  Measured reason
    // (rocprofv3, batch 32): with the fragment-shaped 8-byte V loads every lane of a quad hit a different 128-B row,
    // 40 M L1 accesses per launch for 2.1 M lines -- the kernel was bound by the L1 tag rate, not by HBM.
    constexpr int NP = NT / 2;
    static_assert(NT % 2 == 0, "tiles come in pairs");
    const bool bs_pow2 = (bs & (bs - 1)) == 0;
    const int bs_shift = __ffs(bs) - 1;
    // (round 3 A/B on one box: the table entries requested once per wave ahead of the context length -- one relaxed atomic load,
    // handed out by ds_bpermute as pa_mfma_chunk does -- ran the batch-32 step 1.5 % SLOWER, 5905 vs 5995 tok/s three times over,
    // batch 1 unchanged: the compiler already issues these loads together, and the bpermute sits in front of every K / V address)
    const uint32_t* bt = p.block_tables + (int64_t)b * p.max_blocks;
    auto locate = [&](int tok, int64_t& blk, int& off) {
        const int q = bs_pow2 ? (tok >> bs_shift) : (tok / bs);
        // blk_s >= 0: the partition lies inside ONE block whose id the kernel already fetched with a scalar load next to the context
        // length (round 6: the per-lane table load was a dependent vector round trip in front of every K / V address)
        blk = blk_s >= 0 ? (int64_t)blk_s : (int64_t)bt[q];
        off = tok - q * bs;
    };
    // Q^T fragments: lane (head c, kg) holds Q[head][32j + 8kg .. +8]
    uint4 qf[D32];
#pragma unroll
    for (int j = 0; j < D32; ++j) {
        qf[j] = make_uint4(0, 0, 0, 0);
        if (c < G)
            qf[j] = *reinterpret_cast<const uint4*>(static_cast<const uint16_t*>(p.q) + (int64_t)b * p.q_stride +
                                                    (int64_t)(hk * G + c) * D + 32 * j + 8 * kg);
    }
    // all K and V of the partition are requested before the first MFMA (fp8 cache: raw bytes now, converted to bf16
    // fragments at use -- e4m3 is exact in bf16; the dequantisation scales fold into the logit scale / normalisation)
    typedef typename std::conditional<KV8, uint2, uint4>::type kraw_t;
    typedef typename std::conditional<KV8, uint2, uint4>::type vraw_t;   // 8 tokens of one channel
    kraw_t kf[NT][D32];
    vraw_t vraw[NP][NTD];
    const int krow_tok = 8 * (c >> 2) + (c & 3);                  // token of MFMA row c inside its pair (tile A; B: +4)
#pragma unroll
    for (int it = 0; it < NT; ++it) {
        int tok = t0 + 32 * (it >> 1) + krow_tok + 4 * (it & 1);
        if (tok >= t1) tok = t0;                                  // masked below; any valid address will do
        int64_t blk; int off;
        locate(tok, blk, off);
        if constexpr (KV8) {
            const uint8_t* kb = static_cast<const uint8_t*>(p.kc) + ((blk * p.Hkv + hk) * (D / 16)) * (int64_t)bs * 16 +
                                (int64_t)off * 16 + 8 * (kg & 1);
#pragma unroll
            for (int j = 0; j < D32; ++j) kf[it][j] = *reinterpret_cast<const uint2*>(kb + (int64_t)(2 * j + (kg >> 1)) * bs * 16);
        } else {
            const uint16_t* kb = kc + ((blk * p.Hkv + hk) * (D / 8)) * (int64_t)bs * 8 + (int64_t)off * 8;
#pragma unroll
            for (int j = 0; j < D32; ++j) kf[it][j] = *reinterpret_cast<const uint4*>(kb + (int64_t)(4 * j + kg) * bs * 8);
        }
    }
    const int vq = lane & 3, vch = lane >> 2;                      // load placement: 8-token chunk vq of channel 16nt + vch
#pragma unroll
    for (int ip = 0; ip < NP; ++ip) {
        int tok = t0 + 32 * ip + 8 * vq;
        if (tok >= t1) tok = t0;
        int64_t blk; int off;
        locate(tok, blk, off);                                     // 8 consecutive tokens never straddle a block (bs % 8 == 0)
        if constexpr (KV8) {
            const uint8_t* vb = static_cast<const uint8_t*>(p.vc) + ((blk * p.Hkv + hk) * D + vch) * (int64_t)bs + off;
#pragma unroll
            for (int nt = 0; nt < NTD; ++nt) vraw[ip][nt] = *reinterpret_cast<const uint2*>(vb + (int64_t)(16 * nt) * bs);
        } else {
            const uint16_t* vb = vc + ((blk * p.Hkv + hk) * D + vch) * (int64_t)bs + off;
#pragma unroll
            for (int nt = 0; nt < NTD; ++nt) vraw[ip][nt] = *reinterpret_cast<const uint4*>(vb + (int64_t)(16 * nt) * bs);
        }
    }
    const float qk_scale = KV8 ? p.scale * p.k_scale : p.scale;
    // ---- S^T = K . Q^T : lane (head c, rows 4kg+v); row 4kg+v of tile `it` is token 32*(it/2) + 8kg + 4*(it&1) + v
    float sc[NT][4];
    m = -1e30f;
#pragma unroll
    for (int it = 0; it < NT; ++it) {
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < D32; ++j) {
            uint4 ka;
            if constexpr (KV8) { const uint2 lo = fp8x4_to_bf16x4(kf[it][j].x), hi = fp8x4_to_bf16x4(kf[it][j].y); ka = make_uint4(lo.x, lo.y, hi.x, hi.y); }
            else ka = kf[it][j];
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, ka),
                                                          __builtin_bit_cast(bf16x8_t, qf[j]), acc, 0, 0, 0);
        }
        const int tokb = t0 + 32 * (it >> 1) + 8 * kg + 4 * (it & 1);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            float s = acc[v] * qk_scale;
            if (p.softcap > 0.f) s = tanhf(s / p.softcap) * p.softcap;
            s = (tokb + v < t1) ? s : -1e30f;
            sc[it][v] = s;
            m = fmaxf(m, s);
        }
    }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    lsum = 0.f;
    uint2 pf[NT];
#pragma unroll
    for (int it = 0; it < NT; ++it) {
        const int tokb = t0 + 32 * (it >> 1) + 8 * kg + 4 * (it & 1);
        float pr[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) pr[v] = (tokb + v < t1) ? __expf(sc[it][v] - m) : 0.f;
        pf[it] = make_uint2(cvt_pk_bf16(pr[0], pr[1]), cvt_pk_bf16(pr[2], pr[3]));
        // the normaliser must match what the MFMA sums: the bf16-rounded probabilities
        lsum += (bf16lo_to_f32(pf[it].x) + bf16hi_to_f32(pf[it].x)) + (bf16lo_to_f32(pf[it].y) + bf16hi_to_f32(pf[it].y));
    }
    lsum += __shfl_xor(lsum, 16, 64);
    lsum += __shfl_xor(lsum, 32, 64);
    // ---- O = P . V : lane (channel 16nt + c, heads 4kg+v); contraction index 8kg+e <-> token 32*ip + 8kg + e
#pragma unroll
    for (int nt = 0; nt < NTD; ++nt) o[nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int src_lane4 = 4 * (4 * c + kg);                        // byte address of lane 4c+kg for ds_bpermute
#pragma unroll
    for (int ip = 0; ip < NP; ++ip) {
        const uint4 pa = make_uint4(pf[2 * ip].x, pf[2 * ip].y, pf[2 * ip + 1].x, pf[2 * ip + 1].y);
        const int tk = t0 + 32 * ip + 8 * kg;                     // first token of this lane's 8
        const bool partial = t0 + 32 * ip + 32 > t1;              // pair reaches beyond the context
#pragma unroll
        for (int nt = 0; nt < NTD; ++nt) {
            uint4 vv;
            if constexpr (KV8) {
                const uint32_t r0 = (uint32_t)__builtin_amdgcn_ds_bpermute(src_lane4, (int)vraw[ip][nt].x);
                const uint32_t r1 = (uint32_t)__builtin_amdgcn_ds_bpermute(src_lane4, (int)vraw[ip][nt].y);
                const uint2 lo = fp8x4_to_bf16x4(r0), hi = fp8x4_to_bf16x4(r1);
                vv = make_uint4(lo.x, lo.y, hi.x, hi.y);
            } else {
                vv.x = (uint32_t)__builtin_amdgcn_ds_bpermute(src_lane4, (int)vraw[ip][nt].x);
                vv.y = (uint32_t)__builtin_amdgcn_ds_bpermute(src_lane4, (int)vraw[ip][nt].y);
                vv.z = (uint32_t)__builtin_amdgcn_ds_bpermute(src_lane4, (int)vraw[ip][nt].z);
                vv.w = (uint32_t)__builtin_amdgcn_ds_bpermute(src_lane4, (int)vraw[ip][nt].w);
            }
            if (partial) {                                        // never multiply 0 by unwritten (maybe NaN) V
                if (tk + 0 >= t1) vv.x &= 0xFFFF0000u;
                if (tk + 1 >= t1) vv.x &= 0x0000FFFFu;
                if (tk + 2 >= t1) vv.y &= 0xFFFF0000u;
                if (tk + 3 >= t1) vv.y &= 0x0000FFFFu;
                if (tk + 4 >= t1) vv.z &= 0xFFFF0000u;
                if (tk + 5 >= t1) vv.z &= 0x0000FFFFu;
                if (tk + 6 >= t1) vv.w &= 0xFFFF0000u;
                if (tk + 7 >= t1) vv.w &= 0x0000FFFFu;
            }
            o[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, pa),
                                                            __builtin_bit_cast(bf16x8_t, vv), o[nt], 0, 0, 0);
        }
    }
}

// The same arithmetic as a LOOP over the 32-token pairs of one wave's chunk [t0, t1) (256 / 512 tokens), for launches with many
// sequences and long contexts (batch 32 at 4 k tokens: 32 768 one-partition waves lived 6.4 us each, 55 % of it parked behind the
// block-table -> K/V dependent round trips, the store drain and the arrival ticket, and held 16 KB in flight for a third of that --
// r03_pmc_sq_b32: SQ_WAIT_ANY 0.55 of SQ_WAVE_CYCLES, 6.4 of 16 possible waves resident per CU, 4.2 TB/s).  Here a wave
//   * reads its chunk's block ids ONCE (one table entry per lane, handed out by ds_bpermute),
//   * keeps one group of loads in flight while it computes on the other: V of pair i is requested before the QK^T of pair i, K of
//     pair i + 1 before the P.V of pair i,
//   * carries the running maximum and sum across pairs (online softmax: O and the sum are rescaled by exp(m_old - m_new) -- the
//     column statistic of head 4kg+v comes from lane 4kg+v through ds_bpermute),
//   * and pays Q, the in-workgroup merge, the partial store and the arrival ticket once per chunk instead of once per 32 tokens.
// Same fragment placement, masking rules and outputs (unnormalised O, m, bf16-rounded sum) as pa_mfma_partition.
template <int D32, bool KV8>
__device__ __forceinline__ void pa_mfma_chunk(const PAParams& p, const int b, const int hk, const int t0, const int t1,
                                              const int lane, const int btv, f32x4_t (&o)[2 * D32], float& m, float& lsum) {
    constexpr int D = 32 * D32, NTD = D / 16;
    const int G = p.H / p.Hkv, bs = p.block_size;
    const int c = lane & 15, kg = lane >> 4;
    const uint16_t* kc = static_cast<const uint16_t*>(p.kc);
    const uint16_t* vc = static_cast<const uint16_t*>(p.vc);
    const bool bs_pow2 = (bs & (bs - 1)) == 0;
    const int bs_shift = __ffs(bs) - 1;
    // block ids of the chunk: lane l of `btv` holds table entry t0 / bs + l (t0 is a multiple of the block size; the chunk has at
    // most 64 blocks -- pa_dispatch checks both); only entries of valid tokens are used
    auto locate = [&](int rel, int64_t& blk, int& off) {              // rel = token - t0
        const int q = bs_pow2 ? (rel >> bs_shift) : (rel / bs);
        blk = (int64_t)__builtin_amdgcn_ds_bpermute(4 * q, btv);
        off = rel - q * bs;
    };
    uint4 qf[D32];
#pragma unroll
    for (int j = 0; j < D32; ++j) {
        qf[j] = make_uint4(0, 0, 0, 0);
        if (c < G)
            qf[j] = *reinterpret_cast<const uint4*>(static_cast<const uint16_t*>(p.q) + (int64_t)b * p.q_stride +
                                                    (int64_t)(hk * G + c) * D + 32 * j + 8 * kg);
    }
    typedef typename std::conditional<KV8, uint2, uint4>::type kraw_t;
    typedef typename std::conditional<KV8, uint2, uint4>::type vraw_t;
    kraw_t kf[2][D32];
    vraw_t vraw[NTD];
    const int krow_tok = 8 * (c >> 2) + (c & 3);                      // token of MFMA row c inside its pair (tile A; B: +4)
    const int vq = lane & 3, vch = lane >> 2;                         // V load placement: 8-token chunk vq of channel 16nt + vch
    const int np = (t1 - t0 + 31) >> 5;                               // pairs of this chunk (>= 1: the caller checked t0 < t1)
    // K of pair ip.  Rows at or beyond t1 read the chunk's first token (masked below); tile B's row is 4 tokens further in
    // the SAME block (block size % 8 == 0), allocated whenever tile A's token is valid.  `fetch` false (nothing left to
    // prefetch): every lane reads the same 16 bytes -- one request instead of a predicated load, which would make the
    // compiler drain the whole queue
    auto issue_k = [&](int ip, bool fetch) {
        int rel = 32 * ip + krow_tok;
        if (!fetch || t0 + rel >= t1) rel = 0;
        int64_t blk; int off;
        locate(rel, blk, off);
        const int f = fetch ? 1 : 0;
        if constexpr (KV8) {
            const uint8_t* kb = static_cast<const uint8_t*>(p.kc) + ((blk * p.Hkv + hk) * (D / 16)) * (int64_t)bs * 16 +
                                (int64_t)f * (off * 16 + 8 * (kg & 1));
#pragma unroll
            for (int it = 0; it < 2; ++it)
#pragma unroll
                for (int j = 0; j < D32; ++j)
                    kf[it][j] = *reinterpret_cast<const uint2*>(kb + (int64_t)f * ((int64_t)(2 * j + (kg >> 1)) * bs * 16 + it * 64));
        } else {
            const uint16_t* kb = kc + ((blk * p.Hkv + hk) * (D / 8)) * (int64_t)bs * 8 + (int64_t)f * off * 8;
#pragma unroll
            for (int it = 0; it < 2; ++it)
#pragma unroll
                for (int j = 0; j < D32; ++j)
                    kf[it][j] = *reinterpret_cast<const uint4*>(kb + (int64_t)f * ((int64_t)(4 * j + kg) * bs * 8 + it * 32));
        }
    };
    auto issue_v = [&](int ip) {
        int rel = 32 * ip + 8 * vq;
        if (t0 + rel >= t1) rel = 0;
        int64_t blk; int off;
        locate(rel, blk, off);                                        // 8 consecutive tokens never straddle a block
        if constexpr (KV8) {
            const uint8_t* vb = static_cast<const uint8_t*>(p.vc) + ((blk * p.Hkv + hk) * D + vch) * (int64_t)bs + off;
#pragma unroll
            for (int nt = 0; nt < NTD; ++nt) vraw[nt] = *reinterpret_cast<const uint2*>(vb + (int64_t)(16 * nt) * bs);
        } else {
            const uint16_t* vb = vc + ((blk * p.Hkv + hk) * D + vch) * (int64_t)bs + off;
#pragma unroll
            for (int nt = 0; nt < NTD; ++nt) vraw[nt] = *reinterpret_cast<const uint4*>(vb + (int64_t)(16 * nt) * bs);
        }
    };
    const float qk_scale = KV8 ? p.scale * p.k_scale : p.scale;
    const int src_lane4 = 4 * (4 * c + kg);                           // V: byte address of lane 4c+kg for ds_bpermute
    float m_run = -1e30f, l_run = 0.f;                                // head c's running maximum (all four kg lanes), this lane's share of the sum
#pragma unroll
    for (int nt = 0; nt < NTD; ++nt) o[nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    issue_k(0, true);
    for (int ip = 0; ip < np; ++ip) {
        issue_v(ip);
        // ---- S^T = K . Q^T : lane (head c, rows 4kg+v); row 4kg+v of tile `it` is token 32ip + 8kg + 4it + v
        float sc[2][4];
        float mp = -1e30f;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < D32; ++j) {
                uint4 ka;
                if constexpr (KV8) { const uint2 lo = fp8x4_to_bf16x4(kf[it][j].x), hi = fp8x4_to_bf16x4(kf[it][j].y); ka = make_uint4(lo.x, lo.y, hi.x, hi.y); }
                else ka = kf[it][j];
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, ka),
                                                              __builtin_bit_cast(bf16x8_t, qf[j]), acc, 0, 0, 0);
            }
#pragma unroll
            for (int v = 0; v < 4; ++v) sc[it][v] = acc[v] * qk_scale;
        }
        // K of the next pair goes out as soon as this pair's K registers are free
        issue_k(ip + 1 < np ? ip + 1 : 0, ip + 1 < np);
        if (p.softcap > 0.f) {                                        // one uniform branch for the eight logits
#pragma unroll
            for (int it = 0; it < 2; ++it)
#pragma unroll
                for (int v = 0; v < 4; ++v) sc[it][v] = tanhf(sc[it][v] / p.softcap) * p.softcap;
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int tokb = t0 + 32 * ip + 8 * kg + 4 * it;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                sc[it][v] = (tokb + v < t1) ? sc[it][v] : -1e30f;
                mp = fmaxf(mp, sc[it][v]);
            }
        }
        mp = fmaxf(mp, __shfl_xor(mp, 16, 64));
        mp = fmaxf(mp, __shfl_xor(mp, 32, 64));
        const float m_new = fmaxf(m_run, mp);                         // finite: every pair of the loop has a valid token
        const float alpha = __expf(m_run - m_new);                    // first pair: exp(-1e30 - m) = 0
        m_run = m_new;
        uint2 pf[2];
        float lp = 0.f;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int tokb = t0 + 32 * ip + 8 * kg + 4 * it;
            float pr[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) pr[v] = (tokb + v < t1) ? __expf(sc[it][v] - m_new) : 0.f;
            pf[it] = make_uint2(cvt_pk_bf16(pr[0], pr[1]), cvt_pk_bf16(pr[2], pr[3]));
            // the normaliser must match what the MFMA sums: the bf16-rounded probabilities
            lp += (bf16lo_to_f32(pf[it].x) + bf16hi_to_f32(pf[it].x)) + (bf16lo_to_f32(pf[it].y) + bf16hi_to_f32(pf[it].y));
        }
        l_run = fmaf(l_run, alpha, lp);
        // ---- O = O * alpha + P . V : lane (channel 16nt + c, heads 4kg+v) -- head 4kg+v's alpha sits in lane 4kg+v
        float av[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) av[v] = __int_as_float(__builtin_amdgcn_ds_bpermute(4 * (4 * kg + v), __float_as_int(alpha)));
        const uint4 pa = make_uint4(pf[0].x, pf[0].y, pf[1].x, pf[1].y);
        // never multiply 0 by unwritten (maybe NaN) V: tokens at or beyond t1 are cleared in the B fragment (branch-free: the
        // masks are all ones for every pair but the last)
        const int tk = t0 + 32 * ip + 8 * kg;                         // first token of this lane's 8
        uint32_t vmask[4];
#pragma unroll
        for (int e = 0; e < 4; ++e)
            vmask[e] = (tk + 2 * e < t1 ? 0x0000FFFFu : 0u) | (tk + 2 * e + 1 < t1 ? 0xFFFF0000u : 0u);
#pragma unroll
        for (int nt = 0; nt < NTD; ++nt) {
            uint4 vv;
            if constexpr (KV8) {
                const uint32_t r0 = (uint32_t)__builtin_amdgcn_ds_bpermute(src_lane4, (int)vraw[nt].x);
                const uint32_t r1 = (uint32_t)__builtin_amdgcn_ds_bpermute(src_lane4, (int)vraw[nt].y);
                const uint2 lo = fp8x4_to_bf16x4(r0), hi = fp8x4_to_bf16x4(r1);
                vv = make_uint4(lo.x, lo.y, hi.x, hi.y);
            } else {
                vv.x = (uint32_t)__builtin_amdgcn_ds_bpermute(src_lane4, (int)vraw[nt].x);
                vv.y = (uint32_t)__builtin_amdgcn_ds_bpermute(src_lane4, (int)vraw[nt].y);
                vv.z = (uint32_t)__builtin_amdgcn_ds_bpermute(src_lane4, (int)vraw[nt].z);
                vv.w = (uint32_t)__builtin_amdgcn_ds_bpermute(src_lane4, (int)vraw[nt].w);
            }
            vv.x &= vmask[0]; vv.y &= vmask[1]; vv.z &= vmask[2]; vv.w &= vmask[3];
            f32x4_t on = o[nt];
#pragma unroll
            for (int v = 0; v < 4; ++v) on[v] *= av[v];
            o[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, pa),
                                                            __builtin_bit_cast(bf16x8_t, vv), on, 0, 0, 0);
        }
    }
    m = m_run;
    l_run += __shfl_xor(l_run, 16, 64);
    l_run += __shfl_xor(l_run, 32, 64);
    lsum = l_run;
}

// WPB = waves per workgroup.  1: one partition per workgroup (many sequences: the grid is large anyway).  > 1: the
// workgroup takes WPB consecutive partitions, one per wave, and merges them in LDS before anything goes to global
// memory -- WPB x fewer partials for the reduce / fused merge, which is what bounds the step at batch 1 (129
// partitions per head at 4 k context).
// every field of the parameter block in ONE burst of scalar loads at wave entry (round 6; see qmm_kernarg_burst in qmatmul.hip): the
// compiler loads a kernarg field in the block that first uses it, and this kernel's prologue was context_lens pointer -> wait -> the
// context length -> wait -> arrive pointer -> wait -> the rest -> wait, four dependent scalar round trips in front of the first K request
#ifndef PA_KARG_BURST
#define PA_KARG_BURST 0      // measured neutral on the batch-1 step (11.15 vs 11.2 us per launch: the kernel's chain is context length -> table entry -> K / V -> merge -> ticket, not its kernarg); off
#endif
__device__ __forceinline__ void pa_kernarg_burst(const PAParams& p) {
#if PA_KARG_BURST
    asm volatile("" ::"s"(p.out), "s"(p.tmp_out), "s"(p.max_logits), "s"(p.exp_sums), "s"(p.q), "s"(p.kc), "s"(p.vc), "s"(p.block_tables),
                 "s"(p.context_lens), "s"(p.H), "s"(p.Hkv), "s"(p.D), "s"(p.block_size), "s"(p.max_blocks), "s"(p.partition_size),
                 "s"(p.max_partitions), "s"(p.scale), "s"(p.softcap), "s"(p.q_stride), "s"(p.arrive), "s"(p.kv8), "s"(p.k_scale), "s"(p.v_scale));
#endif
}

// Round 6 (-DMI355_PA_TIMELINE builds only, tools/exp_b1_timeline.py): eight 100 MHz timestamps per wave of the batch-1 attention launch, kept in
// registers, written at exit:  0 entry  1 context length known  2 partition done (table entry -> K / V -> QK^T -> softmax -> P.V)
// 3 state in LDS + barrier  4 workgroup merge done, partial stores issued  5 stores drained + barrier  6 arrival ticket back  7 exit
#ifdef MI355_PA_TIMELINE
__device__ unsigned long long* g_pa_ts = nullptr;
extern "C" int mi355_debug_set_pa_timestamps(void* dev_ptr) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_pa_ts), &dev_ptr, sizeof(dev_ptr)); }
#define PA_TL_DECL unsigned long long ptl[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define PA_TL(i) do { ptl[i] = wall_clock64(); } while (0)
#define PA_TL_FLUSH() do { ptl[7] = wall_clock64(); if ((threadIdx.x & 63) == 0 && g_pa_ts) { \
        unsigned long long* d_ = g_pa_ts + ((size_t)((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 16 + (threadIdx.x >> 6)) * 8; \
        for (int i_ = 0; i_ < 8; ++i_) d_[i_] = ptl[i_]; } } while (0)
#else
#define PA_TL_DECL ((void)0)
#define PA_TL(i) ((void)0)
#define PA_TL_FLUSH() ((void)0)
#endif

template <int D32, int NT, bool KV8 = false, int WPB = 1>
__global__ void __launch_bounds__(64 * WPB) paged_attn_mfma_kernel(const PAParams p) {
    PA_TL_DECL;
    PA_TL(0);
    pa_kernarg_burst(p);
    constexpr int D = 32 * D32, NTD = D / 16;
    const int hk = blockIdx.x, b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // scalar: partition index, table address and every `wave`-dependent branch stay on the scalar unit
    const int part = blockIdx.z * WPB + wave;
    // a context longer than the launch was sized for is truncated to the grid (the host layer refuses such a step): the
    // partition count below must match the launched grid or the fused merge's ticket never completes
    const int t0 = part * p.partition_size;
    // looped chunks: the table entries of this wave's chunk go out before the context length is known -- lane l asks for entry
    // t0 / bs + l (clamped to the row: entries beyond the context hold no block id and are never used).  A relaxed atomic load: a
    // plain one is sunk below the context-length branch by the compiler, back into the dependent chain.
    int btv = 0;
    if constexpr (NT == 0)
        btv = (int)__hip_atomic_load(p.block_tables + (int64_t)b * p.max_blocks + min(t0 / p.block_size + lane, p.max_blocks - 1),
                                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
#ifndef PA_SCALAR_TABLE
#define PA_SCALAR_TABLE 1
#endif
    // one-block partitions (NT tiles = 32 tokens inside a block of 32 k tokens): the block id is wave-uniform -- a SCALAR load through the
    // constant address space, in flight together with the context length (the table was written by an earlier launch / copy)
    int blk_s = -1;
    if constexpr (PA_SCALAR_TABLE && NT == 2) {
        if ((p.block_size & 31) == 0 && p.partition_size == 32) {
            typedef const uint32_t __attribute__((address_space(4))) * cbt_t;
            const int q0 = min(t0 / p.block_size, p.max_blocks - 1);
            blk_s = (int)*reinterpret_cast<cbt_t>(reinterpret_cast<uintptr_t>(p.block_tables + (int64_t)b * p.max_blocks + q0));
        }
    }
    const int ctx = min((int)p.context_lens[b], p.max_partitions * p.partition_size);
    if (blockIdx.z * WPB * p.partition_size >= ctx) return;        // uniform for the workgroup
    PA_TL(1);
    const bool live = t0 < ctx;
    const int t1 = min(ctx, t0 + p.partition_size);
    const int G = p.H / p.Hkv;
    const int c = lane & 15, kg = lane >> 4;
    extern __shared__ __attribute__((aligned(16))) float pa_smem[];   // WPB > 1 only
    f32x4_t o[NTD];
    float m = -1e30f, lsum = 0.f;
#pragma unroll
    for (int nt = 0; nt < NTD; ++nt) o[nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (live) {
        if constexpr (NT == 0) pa_mfma_chunk<D32, KV8>(p, b, hk, t0, t1, lane, btv, o, m, lsum);  // partition = a chunk of 32-token pairs, looped
        else pa_mfma_partition<D32, NT, KV8>(p, b, hk, t0, t1, lane, o, m, lsum, blk_s);
    }
#ifdef MI355_PA_TIMELINE
    asm volatile("" ::"v"(o[0][0]), "v"(lsum));
#endif
    PA_TL(2);
    const int group_tokens = p.partition_size * WPB;               // tokens behind one partial in tmp_out
    const int pslot = blockIdx.z;
    if constexpr (WPB > 1) {
        // ---- in-workgroup merge: [wave][head < G][D] unnormalised O + (m, lsum) per (wave, head)
        float* sm_o = pa_smem;                                     // [WPB][G][D]
        float* sm_m = sm_o + (size_t)WPB * G * D;                  // [WPB][G]
        float* sm_l = sm_m + WPB * G;                              // [WPB][G]
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int head = 4 * kg + v;
            if (head < G) {
#pragma unroll
                for (int nt = 0; nt < NTD; ++nt) sm_o[((size_t)wave * G + head) * D + 16 * nt + c] = o[nt][v];
            }
        }
        if (lane < G) { sm_m[wave * G + lane] = m; sm_l[wave * G + lane] = lsum; }
        __syncthreads();
        PA_TL(3);
        const float vs = KV8 ? p.v_scale : 1.f;
        for (int idx = threadIdx.x; idx < G * D; idx += 64 * WPB) {
            const int g = idx / D, d = idx - g * D;
            float M = -1e30f;
#pragma unroll
            for (int w = 0; w < WPB; ++w) M = fmaxf(M, sm_m[w * G + g]);
            float den = 0.f, val = 0.f;
#pragma unroll
            for (int w = 0; w < WPB; ++w) {
                const float e = __expf(sm_m[w * G + g] - M);
                den = fmaf(sm_l[w * G + g], e, den);
                val = fmaf(sm_o[((size_t)w * G + g) * D + d], e, val);
            }
            const float outv = (den > 0.f ? val / den : 0.f) * vs;
            const int h = hk * G + g;
            if (p.max_partitions > 1) {
                const int64_t pi = ((int64_t)b * p.H + h) * p.max_partitions + pslot;
                if (p.arrive) {
                    __hip_atomic_store(p.tmp_out + pi * D + d, outv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (d == 0) {
                        __hip_atomic_store(p.max_logits + pi, M, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(p.exp_sums + pi, den, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                } else {
                    p.tmp_out[pi * D + d] = outv;
                    if (d == 0) { p.max_logits[pi] = M; p.exp_sums[pi] = den; }
                }
            } else {
                static_cast<uint16_t*>(p.out)[((int64_t)b * p.H + h) * D + d] = f32_to_bf16(outv);
            }
        }
        if (p.max_partitions <= 1 || p.arrive == nullptr) return;
        PA_TL(4);
        // fused merge, workgroup form: every wave drains its write-through stores, one thread takes the ticket, the
        // last workgroup merges with one WAVE per query head (64 lanes x D/64 channels)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        PA_TL(5);
        unsigned* sm_t = reinterpret_cast<unsigned*>(pa_smem);
        if (threadIdx.x == 0)
            sm_t[0] = __hip_atomic_fetch_add(p.arrive + (int64_t)b * p.Hkv + hk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        PA_TL(6);
        const int Pg = (ctx + group_tokens - 1) / group_tokens;
        if ((int)sm_t[0] != Pg - 1) { PA_TL_FLUSH(); return; }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (threadIdx.x == 0) __hip_atomic_store(p.arrive + (int64_t)b * p.Hkv + hk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // one wave per query head; ONE round trip per 32 partials: lane q holds (exp_sum, max_logit) of partial q, every lane its
        // D/64 channels of all 32 -- everything is requested before anything is used (the first form walked the partials 8 at a
        // time, a memory round trip each: 3 us of serial tail behind a 6 us attention at 16 partials per head, 5 us at 32)
        constexpr int DL2 = D / 64, MU2 = 32;
        for (int g = wave; g < G; g += WPB) {
            const int64_t pi0 = ((int64_t)b * p.H + hk * G + g) * p.max_partitions;
            float M = -1e30f;
            float accv[DL2], den = 0.f;
#pragma unroll
            for (int e = 0; e < DL2; ++e) accv[e] = 0.f;
            if (Pg <= 64) {
                const float ml_q = lane < Pg ? p.max_logits[pi0 + lane] : -1e30f;
                const float es_q = lane < Pg ? p.exp_sums[pi0 + lane] : 0.f;
                float tv[MU2][DL2];
#pragma unroll
                for (int u = 0; u < MU2; ++u) {
                    const int q = u < Pg ? u : Pg - 1;
#pragma unroll
                    for (int e = 0; e < DL2; ++e) tv[u][e] = p.tmp_out[(pi0 + q) * D + lane * DL2 + e];
                }
                M = wave_max_dpp(ml_q);
                const float wl = es_q * __expf(ml_q - M);          // 0 beyond Pg (es_q = 0)
                den = wave_sum_dpp(wl);
                const int wbits = __float_as_int(wl);
#pragma unroll
                for (int u = 0; u < MU2; ++u) {
                    const float wq = __int_as_float(__builtin_amdgcn_readlane(wbits, u));
#pragma unroll
                    for (int e = 0; e < DL2; ++e) accv[e] = fmaf(tv[u][e], wq, accv[e]);
                }
                for (int q0 = MU2; q0 < Pg; q0 += MU2) {           // 33..64 partials: one more round trip
#pragma unroll
                    for (int u = 0; u < MU2; ++u) {
                        const int q = q0 + u < Pg ? q0 + u : Pg - 1;
#pragma unroll
                        for (int e = 0; e < DL2; ++e) tv[u][e] = p.tmp_out[(pi0 + q) * D + lane * DL2 + e];
                    }
#pragma unroll
                    for (int u = 0; u < MU2; ++u) {
                        const float wq = __int_as_float(__builtin_amdgcn_readlane(wbits, q0 + u));
#pragma unroll
                        for (int e = 0; e < DL2; ++e) accv[e] = fmaf(tv[u][e], wq, accv[e]);
                    }
                }
            } else {
                for (int q = lane; q < Pg; q += 64) M = fmaxf(M, p.max_logits[pi0 + q]);
                M = wave_max(M);
                constexpr int MU3 = 8;
                for (int q0 = 0; q0 < Pg; q0 += MU3) {
                    float tv[MU3][DL2], es[MU3], ml[MU3];
#pragma unroll
                    for (int u = 0; u < MU3; ++u) {
                        const int q = (q0 + u < Pg) ? q0 + u : Pg - 1;
                        es[u] = p.exp_sums[pi0 + q];
                        ml[u] = p.max_logits[pi0 + q];
#pragma unroll
                        for (int e = 0; e < DL2; ++e) tv[u][e] = p.tmp_out[(pi0 + q) * D + lane * DL2 + e];
                    }
#pragma unroll
                    for (int u = 0; u < MU3; ++u) {
                        const float wq = (q0 + u < Pg) ? es[u] * __expf(ml[u] - M) : 0.f;
                        den += wq;
#pragma unroll
                        for (int e = 0; e < DL2; ++e) accv[e] = fmaf(tv[u][e], wq, accv[e]);
                    }
                }
            }
            const float inv = den > 0.f ? 1.f / den : 0.f;
            uint16_t* op = static_cast<uint16_t*>(p.out) + ((int64_t)b * p.H + hk * G + g) * D + lane * DL2;
#pragma unroll
            for (int e = 0; e < DL2; ++e) op[e] = f32_to_bf16(accv[e] * inv);
        }
        PA_TL_FLUSH();
        return;
    } else {
    // ---- epilogue: rows (heads 4kg+v) need the column statistics of lane (4kg+v)
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int head = 4 * kg + v;
        const float lh = __shfl(lsum, head, 64), mh = __shfl(m, head, 64);
        if (head >= G) continue;
        const int h = hk * G + head;
        const float inv = (lh > 0.f ? 1.f / lh : 0.f) * (KV8 ? p.v_scale : 1.f);
        if (p.max_partitions > 1) {
            const int64_t pi = ((int64_t)b * p.H + h) * p.max_partitions + pslot;
            if (p.arrive) {
                // fused merge: partials go out WRITE-THROUGH (sc1: relaxed agent-scope stores), so the hand-off
                // needs no release fence (a release would write back the whole L2 from every wave)
#pragma unroll
                for (int nt = 0; nt < NTD; ++nt)
                    __hip_atomic_store(p.tmp_out + pi * D + 16 * nt + c, o[nt][v] * inv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (c == 0) {
                    __hip_atomic_store(p.max_logits + pi, mh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(p.exp_sums + pi, lh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            } else {
#pragma unroll
                for (int nt = 0; nt < NTD; ++nt) p.tmp_out[pi * D + 16 * nt + c] = o[nt][v] * inv;
                if (c == 0) { p.max_logits[pi] = mh; p.exp_sums[pi] = lh; }
            }
        } else {
            uint16_t* op = static_cast<uint16_t*>(p.out) + ((int64_t)b * p.H + h) * D;
#pragma unroll
            for (int nt = 0; nt < NTD; ++nt) op[16 * nt + c] = f32_to_bf16(o[nt][v] * inv);
        }
    }
    if (p.max_partitions <= 1 || p.arrive == nullptr) return;
    }

    // ---- fused merge of the partitions (replaces the separate reduce launch): the wave that arrives last for
    // this (sequence, kv head) merges all partials.  Placement-independent hand-off (guide G16, form R1): the
    // payload was stored write-through, every wave drains its stores, then takes an arrival ticket; the last
    // arriver does ONE agent-scope acquire and reads with plain loads.
    if constexpr (WPB == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned ticket = 0;
    if (lane == 0) ticket = __hip_atomic_fetch_add(p.arrive + (int64_t)b * p.Hkv + hk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    ticket = __shfl(ticket, 0, 64);
    const int P = (ctx + group_tokens - 1) / group_tokens;
    if ((int)ticket != P - 1) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (lane == 0) __hip_atomic_store(p.arrive + (int64_t)b * p.Hkv + hk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    constexpr int DL = D / 16;                                     // channels per lane: 16 lanes cover one head
    for (int g0 = 0; g0 < G; g0 += 4) {
        const int g = g0 + kg;                                     // 4 heads in flight, 16 lanes each
        const bool gl = g < G;
        const int64_t pi0 = ((int64_t)b * p.H + hk * G + (gl ? g : 0)) * p.max_partitions;
        float M = -1e30f;
        for (int q = c; q < P; q += 16) M = fmaxf(M, p.max_logits[pi0 + q]);
        M = fmaxf(M, __shfl_xor(M, 8, 64));
        M = fmaxf(M, __shfl_xor(M, 4, 64));
        M = fmaxf(M, __shfl_xor(M, 2, 64));
        M = fmaxf(M, __shfl_xor(M, 1, 64));
        float accv[DL];
#pragma unroll
        for (int e = 0; e < DL; ++e) accv[e] = 0.f;
        float den = 0.f;
        constexpr int MU = 4;                                      // partials in flight per lane
        for (int q0 = 0; q0 < P; q0 += MU) {
            float4 t[MU][DL / 4];
            float es[MU], ml[MU];
#pragma unroll
            for (int u = 0; u < MU; ++u) {
                const int q = (q0 + u < P) ? q0 + u : P - 1;
                es[u] = p.exp_sums[pi0 + q];
                ml[u] = p.max_logits[pi0 + q];
                const float4* tp = reinterpret_cast<const float4*>(p.tmp_out + (pi0 + q) * D + c * DL);
#pragma unroll
                for (int e = 0; e < DL / 4; ++e) t[u][e] = tp[e];
            }
#pragma unroll
            for (int u = 0; u < MU; ++u) {
                const float wq = (q0 + u < P) ? es[u] * __expf(ml[u] - M) : 0.f;
                den += wq;
#pragma unroll
                for (int e = 0; e < DL / 4; ++e) {
                    accv[4 * e + 0] = fmaf(t[u][e].x, wq, accv[4 * e + 0]);
                    accv[4 * e + 1] = fmaf(t[u][e].y, wq, accv[4 * e + 1]);
                    accv[4 * e + 2] = fmaf(t[u][e].z, wq, accv[4 * e + 2]);
                    accv[4 * e + 3] = fmaf(t[u][e].w, wq, accv[4 * e + 3]);
                }
            }
        }
        if (gl) {
            const float inv = den > 0.f ? 1.f / den : 0.f;
            uint16_t* op = static_cast<uint16_t*>(p.out) + ((int64_t)b * p.H + hk * G + g) * D + c * DL;
#pragma unroll
            for (int e = 0; e < DL; ++e) op[e] = f32_to_bf16(accv[e] * inv);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// EXPERIMENT (tuning key 44 = 2; partition sizes 1024 / 2048, head_dim 128, bf16 cache): K / V through LDS by DMA.
// Every register-staged form above tops out at 4.1-4.7 TB/s of K + V at batch 32 (profiles/r03_b32_attention_loop_probe.txt): the
// raw fragments cost 64 VGPRs and 32 ds_bpermute per 32 tokens, and a 32-token step reads HALF of every 128-byte V row.  Here one
// workgroup (4 waves, one per SIMD) walks a chunk in stages of 64 tokens -- with 64-token blocks a stage is two contiguous 16 KiB
// slabs of the cache -- through a ring of R stage buffers:
//   * every wave copies 8 KiB of the stage global -> LDS with `global_load_lds_dwordx4` (waves 0, 1: the 16 channel-group rows of K;
//     waves 2, 3: the 128 channel rows of V), up to R - 1 stages ahead; one counted `s_waitcnt vmcnt` + one barrier per stage;
//   * the DMA's per-lane SOURCE address carries the permutations, the destination is lane-linear: K's 16-byte token slots are
//     stored in MFMA row order (slot 32 ip + 16 it + r <- token 32 ip + 8 (r >> 2) + (r & 3) + 4 it: the A fragment of a tile is 16
//     consecutive slots of a row -- conflict-free ds_read_b128), V's eight 16-byte slots of a channel row are XOR-swizzled with
//     (channel >> 1) & 7 (a b128 lane group then covers 16 different bank quads);
//   * every wave computes the scores of ALL 64 tokens (16 MFMAs, identical in the four waves: the softmax statistics agree bit
//     for bit without an exchange) and the P.V of ITS 32 channels (4 MFMAs; the B fragment is one ds_read_b128 of a V row);
//   * online softmax across stages as in pa_mfma_chunk; partials go out once per chunk and are merged by the reduce launch.
// Token order inside a 32-token pair, masks and outputs are those of pa_mfma_partition.
template <int R>
__global__ void __launch_bounds__(256, 1) paged_attn_lds_kernel(const PAParams p) {
    constexpr int D = 128, D32 = 4;
    constexpr uint32_t STAGE_B = 32768u;                              // K 16 KiB | V 16 KiB
    extern __shared__ __attribute__((aligned(1024))) uint8_t pal_smem[];
    const int hk = blockIdx.x, b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int c = lane & 15, kg = lane >> 4;
    const int G = p.H / p.Hkv, bs = p.block_size;
    const int t0 = blockIdx.z * p.partition_size;
    const int ctx = min((int)p.context_lens[b], p.max_partitions * p.partition_size);
    if (t0 >= ctx) return;                                            // uniform for the workgroup
    const int t1 = min(ctx, t0 + p.partition_size);
    const int ns = (t1 - t0 + 63) >> 6;                               // stages of this chunk
    const int bs_shift = __ffs(bs) - 1;                               // block size 16 / 32 / 64 (pa_dispatch)
    // table entries of the chunk (at most 64: pa_dispatch), one per lane; only entries of valid tokens are used
    const int nblk = (t1 - t0 + bs - 1) >> bs_shift;
    const int btv = (int)p.block_tables[(int64_t)b * p.max_blocks + (t0 >> bs_shift) + min(lane, nblk - 1)];
    uint4 qf[D32];
#pragma unroll
    for (int j = 0; j < D32; ++j) {
        const int hq = hk * G + (c < G ? c : 0);
        qf[j] = *reinterpret_cast<const uint4*>(static_cast<const uint16_t*>(p.q) + (int64_t)b * p.q_stride + (int64_t)hq * D + 32 * j + 8 * kg);
        if (c >= G) qf[j] = make_uint4(0, 0, 0, 0);
    }
    // the compiler's own wait for these loads must come BEFORE the first DMA goes out: it knows nothing of the DMA queue, and a
    // wait placed later would drain it
    {
        int sink = btv;
        asm volatile("" : "+v"(sink), "+v"(qf[0].x), "+v"(qf[1].x), "+v"(qf[2].x), "+v"(qf[3].x));
    }
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(__attribute__((address_space(3))) void*)pal_smem);
    const uint8_t* kc8 = static_cast<const uint8_t*>(p.kc);
    const uint8_t* vc8 = static_cast<const uint8_t*>(p.vc);
    // ---- this wave's share of a stage: pieces 8 wave .. 8 wave + 7 (1 KiB each; 0..15 K, 16..31 V)
    const int ktok = pal_dma_token(0, lane);                          // token of K slot `lane` (the same in all 16 rows)
    auto issue = [&](int st) {
        const uint32_t dst = lds0 + (uint32_t)(st % R) * STAGE_B + (uint32_t)wave * 8192u;
        const int rel0 = 64 * st;                                     // first token of the stage, relative to t0 (valid: st < ns)
        if (wave < 2) {
            int rel = rel0 + ktok;
            if (t0 + rel >= t1) rel = rel0;
            const int q = rel >> bs_shift;
            const int64_t blk = (int64_t)__builtin_amdgcn_ds_bpermute(4 * q, btv);
            const int off = rel - (q << bs_shift);
            // K [NB][Hkv][16 channel groups][bs][16 B]
            const uint8_t* src = kc8 + ((blk * p.Hkv + hk) * 16 + 8 * wave) * (int64_t)bs * 16 + (int64_t)off * 16;
#pragma unroll
            for (int i = 0; i < 8; ++i) pa_dma16(src + (int64_t)i * bs * 16, dst + (uint32_t)i * 1024u);
        } else {
            // V [NB][Hkv][128 channels][bs][2 B]: piece i of this wave = piece q = 8 wave + i of the stage, channels 8 (q - 16) + (lane >> 3);
            // LDS slot lane & 7 of a row takes the row's slot (lane & 7) ^ ((channel >> 1) & 7), which depends on i through its
            // parity only: two source bases, then 8 channels (8 bs 2 bytes) further per piece
            const uint8_t* srcp[2];
#pragma unroll
            for (int par = 0; par < 2; ++par) {
                int rel = rel0 + pal_dma_token(8 * wave + par, lane);
                if (t0 + rel >= t1) rel = rel0;
                const int q = rel >> bs_shift;
                const int64_t blk = (int64_t)__builtin_amdgcn_ds_bpermute(4 * q, btv);
                const int off = rel - (q << bs_shift);
                srcp[par] = vc8 + (((blk * p.Hkv + hk) * D + pal_dma_row(8 * wave + par, lane)) * (int64_t)bs + off) * 2;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) pa_dma16(srcp[i & 1] + (int64_t)(8 * (i & ~1)) * bs * 2, dst + (uint32_t)i * 1024u);
        }
    };
    const float qk_scale = p.scale;
    float m_run = -1e30f, l_run = 0.f;
    f32x4_t o[2];
    o[0] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    o[1] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int st = 0; st < R - 1; ++st)
        if (st < ns) issue(st);
    for (int i = 0; i < ns; ++i) {
        // stage i must have landed; younger than it in this wave's queue: the stages issued after it (8 DMA instructions each)
        {
            const int ahead = min(ns, i + R - 1) - (i + 1);
            if (ahead >= 3) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
            else if (ahead == 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else if (ahead == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();                                              // everyone's share of stage i; everyone is done with stage i - 1
        if (i + R - 1 < ns) issue(i + R - 1);                         // into the buffer stage i - 1 used
        const uint8_t* Kb = pal_smem + (size_t)(i % R) * STAGE_B;    // the stage: K at 0, V at 16384 (pal_*_read_off)
        // ---- S^T = K . Q^T for the 64 tokens: tile (ip, it), lane (head c, rows 4kg+v) <-> token 32 ip + 8 kg + 4 it + v
        float sc[2][2][4];
        float mp = -1e30f;
#pragma unroll
        for (int ip = 0; ip < 2; ++ip)
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                uint4 ka[D32];
#pragma unroll
                for (int j = 0; j < D32; ++j)
                    ka[j] = *reinterpret_cast<const uint4*>(Kb + pal_k_read_off(j, kg, ip, it, c));
                f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < D32; ++j)
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, ka[j]), __builtin_bit_cast(bf16x8_t, qf[j]), acc, 0, 0, 0);
#pragma unroll
                for (int v = 0; v < 4; ++v) sc[ip][it][v] = acc[v] * qk_scale;
            }
        if (p.softcap > 0.f) {
#pragma unroll
            for (int ip = 0; ip < 2; ++ip)
#pragma unroll
                for (int it = 0; it < 2; ++it)
#pragma unroll
                    for (int v = 0; v < 4; ++v) sc[ip][it][v] = tanhf(sc[ip][it][v] / p.softcap) * p.softcap;
        }
        const int tb = t0 + 64 * i;
#pragma unroll
        for (int ip = 0; ip < 2; ++ip)
#pragma unroll
            for (int it = 0; it < 2; ++it)
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    sc[ip][it][v] = (tb + 32 * ip + 8 * kg + 4 * it + v < t1) ? sc[ip][it][v] : -1e30f;
                    mp = fmaxf(mp, sc[ip][it][v]);
                }
        mp = fmaxf(mp, __shfl_xor(mp, 16, 64));
        mp = fmaxf(mp, __shfl_xor(mp, 32, 64));
        const float m_new = fmaxf(m_run, mp);                         // finite: every stage of the loop has a valid token
        const float alpha = __expf(m_run - m_new);
        m_run = m_new;
        uint4 pa[2];
        float lp = 0.f;
#pragma unroll
        for (int ip = 0; ip < 2; ++ip) {
            uint32_t w[4];
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                float pr[4];
#pragma unroll
                for (int v = 0; v < 4; ++v) pr[v] = (tb + 32 * ip + 8 * kg + 4 * it + v < t1) ? __expf(sc[ip][it][v] - m_new) : 0.f;
                w[2 * it] = cvt_pk_bf16(pr[0], pr[1]);
                w[2 * it + 1] = cvt_pk_bf16(pr[2], pr[3]);
                // the normaliser must match what the MFMA sums: the bf16-rounded probabilities
                lp += (bf16lo_to_f32(w[2 * it]) + bf16hi_to_f32(w[2 * it])) + (bf16lo_to_f32(w[2 * it + 1]) + bf16hi_to_f32(w[2 * it + 1]));
            }
            pa[ip] = make_uint4(w[0], w[1], w[2], w[3]);              // tokens 32 ip + 8 kg .. + 7 of head c
        }
        l_run = fmaf(l_run, alpha, lp);
        // ---- O = O * alpha + P . V for this wave's channels 32 wave .. 32 wave + 31: lane (channel 16 nt + c, heads 4kg+v)
        float av[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) av[v] = __int_as_float(__builtin_amdgcn_ds_bpermute(4 * (4 * kg + v), __float_as_int(alpha)));
#pragma unroll
        for (int n2 = 0; n2 < 2; ++n2) {
            const int ch = 32 * wave + 16 * n2 + c;
            f32x4_t on = o[n2];
#pragma unroll
            for (int v = 0; v < 4; ++v) on[v] *= av[v];
#pragma unroll
            for (int ip = 0; ip < 2; ++ip) {
                uint4 vv = *reinterpret_cast<const uint4*>(Kb + pal_v_read_off(ch, ip, kg));
                // never multiply 0 by unwritten (maybe NaN) V: tokens at or beyond t1 are cleared
                const int tk = tb + 32 * ip + 8 * kg;
                vv.x &= (tk + 0 < t1 ? 0x0000FFFFu : 0u) | (tk + 1 < t1 ? 0xFFFF0000u : 0u);
                vv.y &= (tk + 2 < t1 ? 0x0000FFFFu : 0u) | (tk + 3 < t1 ? 0xFFFF0000u : 0u);
                vv.z &= (tk + 4 < t1 ? 0x0000FFFFu : 0u) | (tk + 5 < t1 ? 0xFFFF0000u : 0u);
                vv.w &= (tk + 6 < t1 ? 0x0000FFFFu : 0u) | (tk + 7 < t1 ? 0xFFFF0000u : 0u);
                on = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, pa[ip]), __builtin_bit_cast(bf16x8_t, vv), on, 0, 0, 0);
            }
            o[n2] = on;
        }
    }
    l_run += __shfl_xor(l_run, 16, 64);
    l_run += __shfl_xor(l_run, 32, 64);
    // ---- this wave's 32 channels of every head: rows (heads 4kg+v) need the column statistics of lane (4kg+v)
    const int pslot = blockIdx.z;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int head = 4 * kg + v;
        const float lh = __shfl(l_run, head, 64), mh = __shfl(m_run, head, 64);
        if (head >= G) continue;
        const int h = hk * G + head;
        const float inv = lh > 0.f ? 1.f / lh : 0.f;
        if (p.max_partitions > 1) {
            const int64_t pi = ((int64_t)b * p.H + h) * p.max_partitions + pslot;
#pragma unroll
            for (int n2 = 0; n2 < 2; ++n2) p.tmp_out[pi * D + 32 * wave + 16 * n2 + c] = o[n2][v] * inv;
            if (wave == 0 && c == 0) { p.max_logits[pi] = mh; p.exp_sums[pi] = lh; }
        } else {
            uint16_t* op = static_cast<uint16_t*>(p.out) + ((int64_t)b * p.H + h) * D;
#pragma unroll
            for (int n2 = 0; n2 < 2; ++n2) op[32 * wave + 16 * n2 + c] = f32_to_bf16(o[n2][v] * inv);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Partition size 64 at >= 64 (sequence, kv head) pairs (round 4 default; validated by the whole GPU suite): the LDS-DMA stages of
// paged_attn_lds_kernel as ONE balanced stream per workgroup.
// Ragged batches starve the chunked form (670 chunks of 1024 tokens on 256 CUs: 2.6 rounds, every chunk pays its ring fill).  Here the
// (sequence, 64-token stage) pairs of a kv head form one flat index space of S stages; workgroup w of the W per kv head (W x Hkv = one
// per CU) takes the stages [S w / W, S (w + 1) / W) -- equal shares whatever the context lengths -- and walks them through ONE ring:
// the DMA of the next sequence's first stages is in flight while the last stages of the current one are multiplied.  Every
// (sequence, workgroup) pair that meets leaves a partial (normalised O, max, sum) in slot w; paged_attn_stream_reduce_kernel recomputes
// the cuts and merges the workgroups that touched its sequence.
//   * stages per sequence and their prefix sums live in a wave's lanes (lane = sequence, <= 64 sequences): locating a flat index is a
//     ballot + popcount, no planning launch;
//   * block-table entries are fetched by SCALAR loads (wave-uniform addresses) one stage ahead: they count on lgkmcnt, the vector
//     memory counter stays the DMA's own (the partial stores that enter it only make a counted wait stricter: the wait counts DMA
//     instructions alone);
//   * Q of the (up to 4) sequences of the share is staged in LDS before the first DMA goes out.
#include "pa_stream_cut.h"

// PAS_TS (build-time A/B): 1 = the waves of a workgroup split the TOKENS of a stage (round 4), 0 = they split the output channels
#ifndef PAS_TS
#define PAS_TS 1
#endif
#ifndef PAS_RING
#define PAS_RING 3
#endif
// KV8 (`--kvcache-dtype fp8`, token split only): the cache holds e4m3fn bytes -- K [NB][Hkv][D/16][bs][16], V [NB][Hkv][D][bs] -- so a stage is
// 16 KiB: K [8 channel groups][64 slots of 16 B = one token's 16 channels], V [128 channels][4 slots of 16 B = 16 tokens] (slot s of channel
// ch holds row slot s ^ ((ch >> 2) & 3): the 4-byte fragment reads of a wave instruction hit each of their 32 reachable banks twice -- the
// floor of this placement, pa_lds_layout.h / tests/test_cpu_pa_lds_layout.py).  Same DMA engine, half the
// requests; the lane's bytes become bf16 (exact: 3 mantissa bits) on their way into the MFMA operands; scores carry k_scale, the
// partial's output v_scale (attention.rs:896, is_fp8_keys).
template <int R, bool TS = false, bool KV8 = false>
__global__ void __launch_bounds__(256, 1) paged_attn_stream_kernel(const PAParams p, const int B, const uint32_t* __restrict__ btab,
                                                                    const uint32_t* __restrict__ clens) {
    // (btab / clens = p.block_tables / p.context_lens once more, as __restrict__ kernel arguments: only then does the compiler know that
    // the partial stores cannot clobber them and fetches the wave-uniform table entries with s_load -- through the struct it used
    // vector loads, whose waits drained the DMA queue every stage)
    constexpr int D = 128, D32 = 4, QSEG = 4;
    static_assert(!KV8 || TS, "the fp8 cache takes the token split");
    constexpr uint32_t STAGE_B = KV8 ? 16384u : 32768u;
    constexpr int NDMA = KV8 ? 4 : 8;                                 // DMA instructions per wave and stage
    extern __shared__ __attribute__((aligned(1024))) uint8_t pas_smem[];   // ring [R][stage] | Q [QSEG][16 heads][128] bf16 (| KV8: 32 KiB merge scratch)
    const int w = blockIdx.x, W = gridDim.x, hk = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int c = lane & 15, kg = lane >> 4;
    const int G = p.H / p.Hkv, bs = p.block_size;
    const int bs_shift = __ffs(bs) - 1, E = 64 >> bs_shift;           // block size 16 / 32 / 64: 4 / 2 / 1 table entries per stage
    // ---- lane = sequence: context length, stages, inclusive prefix of the stages
    const int ctx_l = lane < B ? (int)clens[lane] : 0;
    const int n_l = (ctx_l + 63) >> 6;
    int pend = n_l;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(pend, o, 64);
        if (lane >= o) pend += t;
    }
    const int S = __builtin_amdgcn_readlane(pend, 63);
    const int f_lo0 = (int)pas_cut(S, W, w), f_hi0 = (int)pas_cut(S, W, w + 1);
    if (f_lo0 >= f_hi0) return;                                       // uniform for the workgroup
    // sequence of flat stage f = number of sequences that end at or before f (the prefix is non-decreasing)
    auto seq_of = [&](int f) { return __builtin_amdgcn_readfirstlane((int)__popcll(__ballot(pend <= f))); };
    uint8_t* qlds = pas_smem + (size_t)R * STAGE_B;
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(__attribute__((address_space(3))) void*)pas_smem);
    const uint8_t* kc8 = static_cast<const uint8_t*>(p.kc);
    const uint8_t* vc8 = static_cast<const uint8_t*>(p.vc);
    const int ktok = pal_dma_token(0, lane);
    // The share is walked in runs of at most QSEG sequences (one run unless the batch is made of very short sequences): a run stages
    // its Q rows in LDS and fills the ring anew -- inside a run there is no compiler-visible vector load, so no wait of the compiler's
    // ever touches the DMA queue.
    for (int f_lo = f_lo0; f_lo < f_hi0;) {
    const int b0 = seq_of(f_lo);
    const int f_hi = min(f_hi0, __builtin_amdgcn_readlane(pend, min(b0 + QSEG - 1, 63)));
    const int ns = f_hi - f_lo;
    // ---- Q of the run's sequences -> LDS (before any DMA of the run goes out)
    {
        __syncthreads();                                              // the previous run is done with the Q rows and the ring
        const int b_last = seq_of(f_hi - 1);
        const int nq = min(b_last - b0 + 1, QSEG);
        const int row = (int)threadIdx.x >> 4, gs = (int)threadIdx.x & 15;
        for (int sg = 0; sg < nq; ++sg) {
            if (row < G) {
                const uint4 v = *reinterpret_cast<const uint4*>(static_cast<const uint16_t*>(p.q) + (int64_t)(b0 + sg) * p.q_stride +
                                                                (int64_t)(hk * G + row) * D + 8 * gs);
                *reinterpret_cast<uint4*>(qlds + ((size_t)(sg * 16 + row) * D + 8 * gs) * 2) = v;
            }
        }
        __syncthreads();
    }
    // table entries of stage (b, s): E consecutive entries, clamped to the sequence's last block (tokens beyond the context are masked)
    auto load_ent = [&](int f, int (&ent)[4]) {
        const int bq = seq_of(f);
        const int sq = f - (__builtin_amdgcn_readlane(pend, bq) - __builtin_amdgcn_readlane(n_l, bq));
        const int nblk = (__builtin_amdgcn_readlane(ctx_l, bq) + bs - 1) >> bs_shift;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int idx = min(sq * E + min(e, E - 1), nblk - 1);
            ent[e] = (int)btab[(int64_t)bq * p.max_blocks + idx];
        }
    };
    // this wave's share of flat stage f (ring slot i % R): as paged_attn_lds_kernel, the block ids come from `ent`
    auto issue = [&](int i, int f, const int (&ent)[4]) {
        const int bq = seq_of(f);
        const int sq = f - (__builtin_amdgcn_readlane(pend, bq) - __builtin_amdgcn_readlane(n_l, bq));
        const int t1q = __builtin_amdgcn_readlane(ctx_l, bq);
        const uint32_t dst = lds0 + (uint32_t)(i % R) * STAGE_B + (uint32_t)wave * (STAGE_B / 4);
        auto block_of = [&](int tok) {                                // tok: token of the stage (0..63)
            const int e = tok >> bs_shift;
            return (int64_t)(e == 0 ? ent[0] : (e == 1 ? ent[1] : (e == 2 ? ent[2] : ent[3])));
        };
        if constexpr (KV8) {
            if (wave < 2) {                                           // K: channel groups 4 wave .. 4 wave + 3, lane = slot
                int tok = ktok;
                if (64 * sq + tok >= t1q) tok = 0;
                const int64_t blk = block_of(tok);
                const int off = tok & (bs - 1);
                const uint8_t* src = kc8 + ((blk * p.Hkv + hk) * 8 + 4 * wave) * (int64_t)bs * 16 + (int64_t)off * 16;
#pragma unroll
                for (int q = 0; q < 4; ++q) pa_dma16(src + (int64_t)q * bs * 16, dst + (uint32_t)q * 1024u);
            } else {                                                  // V: piece q = channels 16 q' + (lane >> 2), LDS slot lane & 3
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int ch = pal8_dma_row(8 + 4 * (wave - 2) + q, lane);
                    int tok = pal8_dma_token(8 + 4 * (wave - 2) + q, lane);
                    if (64 * sq + tok >= t1q) tok = 0;
                    const int64_t blk = block_of(tok);
                    const int off = tok & (bs - 1);
                    pa_dma16(vc8 + ((blk * p.Hkv + hk) * D + ch) * (int64_t)bs + off, dst + (uint32_t)q * 1024u);
                }
            }
        } else
        if (wave < 2) {
            int tok = ktok;
            if (64 * sq + tok >= t1q) tok = 0;
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
                if (64 * sq + tok >= t1q) tok = 0;
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
        if (st < ns) { load_ent(f_lo + st, ent); issue(st, f_lo + st, ent); }
    if (R - 1 < ns) load_ent(f_lo + R - 1, ent);                      // for the issue of the first iteration
    const float qk_scale = KV8 ? p.scale * p.k_scale : p.scale;
    float m_run = -1e30f, l_run = 0.f;
    constexpr int NO = TS ? 8 : 2;                                    // output tiles of 16 channels per wave
    f32x4_t o[NO];
    uint4 qf[D32];
    int b = -1, pst = 0, t1 = 0, pe = 0;                              // current sequence, its first flat stage, context length, end
    for (int i = 0; i < ns; ++i) {
        const int f = f_lo + i;
        {
            const int ahead = min(ns, i + R - 1) - (i + 1);
            if (ahead >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NDMA) : "memory");
            else if (ahead == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NDMA) : "memory");
            else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();                                              // everyone's share of stage i; everyone is done with stage i - 1
        if (i + R - 1 < ns) {
            issue(i + R - 1, f + R - 1, ent);
            if (i + R < ns) load_ent(f + R, ent);                     // scalar loads: in flight until the next iteration's issue
        }
        if (b < 0 || f >= pe) {                                       // first stage of a sequence (of this share)
            b = seq_of(f);
            pe = __builtin_amdgcn_readlane(pend, b);
            pst = pe - __builtin_amdgcn_readlane(n_l, b);
            t1 = __builtin_amdgcn_readlane(ctx_l, b);
            m_run = -1e30f; l_run = 0.f;
#pragma unroll
            for (int n2 = 0; n2 < NO; ++n2) o[n2] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            const int sg = b - b0;
#pragma unroll
            for (int j = 0; j < D32; ++j) {
                qf[j] = *reinterpret_cast<const uint4*>(qlds + ((size_t)(sg * 16 + (c < G ? c : 0)) * D + 32 * j + 8 * kg) * 2);
                if (c >= G) qf[j] = make_uint4(0, 0, 0, 0);
            }
        }
        const int tb = 64 * (f - pst);                                // first token of the stage inside its sequence
        const uint8_t* Kb = pas_smem + (size_t)(i % R) * STAGE_B;
        if constexpr (TS) {
            // ---- TOKEN SPLIT (round 4): wave w takes the 16 tokens of tile (ip, it) = (w >> 1, w & 1) of the stage for ALL 128 channels with
            // its own running softmax state -- 4 QK MFMAs and 4 scores per lane instead of 16 and 16 in every one of the four waves (the
            // channel split computes the whole stage's scores and probabilities in each wave: four times redundant), P.V as eight K = 16
            // MFMAs on the lane's own four probabilities; the four states are merged through LDS once per (sequence, share).
            const int ip = wave >> 1, it = wave & 1;
            uint4 ka[D32];
            if constexpr (KV8) {
                // channels 32 j + 8 kg .. + 7 of the tile's token c: half (kg & 1) of the 16-byte slot of channel group 2 j + (kg >> 1)
                uint2 k8[D32];
#pragma unroll
                for (int j = 0; j < D32; ++j)
                    k8[j] = *reinterpret_cast<const uint2*>(Kb + pal8_k_read_off(j, kg, ip, it, c));
#pragma unroll
                for (int j = 0; j < D32; ++j) {
                    const uint2 lo = fp8x4_to_bf16x4(k8[j].x), hi = fp8x4_to_bf16x4(k8[j].y);
                    ka[j] = make_uint4(lo.x, lo.y, hi.x, hi.y);
                }
            } else {
#pragma unroll
            for (int j = 0; j < D32; ++j) ka[j] = *reinterpret_cast<const uint4*>(Kb + pal_k_read_off(j, kg, ip, it, c));
            }
            f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < D32; ++j)
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, ka[j]), __builtin_bit_cast(bf16x8_t, qf[j]), acc, 0, 0, 0);
            const int tk = tb + 32 * ip + 8 * kg + 4 * it;           // the lane's four tokens: tk .. tk + 3
            float sc[4], mp = -1e30f;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                sc[v] = acc[v] * qk_scale;
                if (p.softcap > 0.f) sc[v] = tanhf(sc[v] / p.softcap) * p.softcap;
                sc[v] = (tk + v < t1) ? sc[v] : -1e30f;
                mp = fmaxf(mp, sc[v]);
            }
            mp = fmaxf(mp, __shfl_xor(mp, 16, 64));
            mp = fmaxf(mp, __shfl_xor(mp, 32, 64));
            const float m_new = fmaxf(m_run, mp);
            const float alpha = __expf(m_run - m_new);
            m_run = m_new;
            float pr[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) pr[v] = (tk + v < t1) ? __expf(sc[v] - m_new) : 0.f;
            const uint32_t pw0 = cvt_pk_bf16(pr[0], pr[1]), pw1 = cvt_pk_bf16(pr[2], pr[3]);
            const float lp = (bf16lo_to_f32(pw0) + bf16hi_to_f32(pw0)) + (bf16lo_to_f32(pw1) + bf16hi_to_f32(pw1));
            l_run = fmaf(l_run, alpha, lp);
            const uint2 pa2 = make_uint2(pw0, pw1);
            float av[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) av[v] = __int_as_float(__builtin_amdgcn_ds_bpermute(4 * (4 * kg + v), __float_as_int(alpha)));
            const uint32_t vm0 = (tk + 0 < t1 ? 0x0000FFFFu : 0u) | (tk + 1 < t1 ? 0xFFFF0000u : 0u);
            const uint32_t vm1 = (tk + 2 < t1 ? 0x0000FFFFu : 0u) | (tk + 3 < t1 ? 0xFFFF0000u : 0u);
#pragma unroll
            for (int n2 = 0; n2 < 8; ++n2) {
                const int ch = 16 * n2 + c;
                f32x4_t on = o[n2];
#pragma unroll
                for (int v = 0; v < 4; ++v) on[v] *= av[v];
                uint2 vv;
                if constexpr (KV8)                                    // 4 bytes = the lane's 4 tokens: row slot 2 ip + (kg >> 1), byte 8 (kg & 1) + 4 it
                    vv = fp8x4_to_bf16x4(*reinterpret_cast<const uint32_t*>(Kb + pal8_v_read_off(ch, ip, kg, it)));
                else
                    vv = *reinterpret_cast<const uint2*>(Kb + pal_v_read_off(ch, ip, kg) + 8 * it);   // the lane's 4 tokens of channel ch
                vv.x &= vm0; vv.y &= vm1;
                o[n2] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4_t, pa2), __builtin_bit_cast(s16x4_t, vv), on, 0, 0, 0);
            }
            if (f + 1 >= pe || i + 1 >= ns) {
                // ---- last stage of this sequence in this share: merge the four waves' states, leave the partial of (sequence b, workgroup w).
                // Scratch = this stage's ring slot (the next DMA into it goes out behind the next iteration's barrier).
                float lt = l_run + __shfl_xor(l_run, 16, 64);
                lt += __shfl_xor(lt, 32, 64);
                __syncthreads();                                      // every wave is done reading the slot
                // (the four waves' (m, l) pairs used to sit in front of `ob` in the slot -- at G = 16 the slot's 32 KiB are
                // exactly ob's 4 x 16 x 128 floats, and the last 512 B ran into the next slot's K / V in flight; they have their own array now)
                __shared__ float pas_stt[128];                        // [4 waves][16 heads][m, l]
                float* stt = pas_stt;
                // (KV8: a stage is 16 KiB -- the scratch has its own 32 KiB behind the Q rows)
                float* ob = reinterpret_cast<float*>((KV8 && G > 8) ? pas_smem + (size_t)R * STAGE_B + 16384 : pas_smem + (size_t)(i % R) * STAGE_B);   // [4 waves][G heads][128 channels]: <= 32 KiB at G <= 16
                if (kg == 0) { stt[(wave * 16 + c) * 2] = m_run; stt[(wave * 16 + c) * 2 + 1] = lt; }
                __syncthreads();
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int head = 4 * kg + v;
                    if (head < G) {
                        float M = stt[head * 2];
#pragma unroll
                        for (int ww = 1; ww < 4; ++ww) M = fmaxf(M, stt[(ww * 16 + head) * 2]);
                        const float fw = __expf(stt[(wave * 16 + head) * 2] - M);
#pragma unroll
                        for (int n2 = 0; n2 < 8; ++n2) ob[(size_t)(wave * G + head) * D + 16 * n2 + c] = o[n2][v] * fw;
                    }
                }
                __syncthreads();
                for (int idx = (int)threadIdx.x; idx < G * D; idx += 256) {
                    const int head = idx >> 7, ch = idx & (D - 1);
                    float M = stt[head * 2];
#pragma unroll
                    for (int ww = 1; ww < 4; ++ww) M = fmaxf(M, stt[(ww * 16 + head) * 2]);
                    float L = 0.f, sum = 0.f;
#pragma unroll
                    for (int ww = 0; ww < 4; ++ww) {
                        L = fmaf(stt[(ww * 16 + head) * 2 + 1], __expf(stt[(ww * 16 + head) * 2] - M), L);
                        sum += ob[(size_t)(ww * G + head) * D + ch];
                    }
                    const int64_t pi = ((int64_t)b * p.H + hk * G + head) * p.max_partitions + w;
                    p.tmp_out[pi * D + ch] = sum * (L > 0.f ? 1.f / L : 0.f) * (KV8 ? p.v_scale : 1.f);
                    if (ch == 0) { p.max_logits[pi] = M; p.exp_sums[pi] = L; }
                }
            }
        } else {
            float sc[2][2][4];
            float mp = -1e30f;
    #pragma unroll
            for (int ip = 0; ip < 2; ++ip)
    #pragma unroll
                for (int it = 0; it < 2; ++it) {
                    uint4 ka[D32];
    #pragma unroll
                    for (int j = 0; j < D32; ++j) ka[j] = *reinterpret_cast<const uint4*>(Kb + pal_k_read_off(j, kg, ip, it, c));
                    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    #pragma unroll
                    for (int j = 0; j < D32; ++j)
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, ka[j]), __builtin_bit_cast(bf16x8_t, qf[j]), acc, 0, 0, 0);
    #pragma unroll
                    for (int v = 0; v < 4; ++v) sc[ip][it][v] = acc[v] * qk_scale;
                }
            if (p.softcap > 0.f) {
    #pragma unroll
                for (int ip = 0; ip < 2; ++ip)
    #pragma unroll
                    for (int it = 0; it < 2; ++it)
    #pragma unroll
                        for (int v = 0; v < 4; ++v) sc[ip][it][v] = tanhf(sc[ip][it][v] / p.softcap) * p.softcap;
            }
    #pragma unroll
            for (int ip = 0; ip < 2; ++ip)
    #pragma unroll
                for (int it = 0; it < 2; ++it)
    #pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        sc[ip][it][v] = (tb + 32 * ip + 8 * kg + 4 * it + v < t1) ? sc[ip][it][v] : -1e30f;
                        mp = fmaxf(mp, sc[ip][it][v]);
                    }
            mp = fmaxf(mp, __shfl_xor(mp, 16, 64));
            mp = fmaxf(mp, __shfl_xor(mp, 32, 64));
            const float m_new = fmaxf(m_run, mp);
            const float alpha = __expf(m_run - m_new);
            m_run = m_new;
            uint4 pa[2];
            float lp = 0.f;
    #pragma unroll
            for (int ip = 0; ip < 2; ++ip) {
                uint32_t pw[4];
    #pragma unroll
                for (int it = 0; it < 2; ++it) {
                    float pr[4];
    #pragma unroll
                    for (int v = 0; v < 4; ++v) pr[v] = (tb + 32 * ip + 8 * kg + 4 * it + v < t1) ? __expf(sc[ip][it][v] - m_new) : 0.f;
                    pw[2 * it] = cvt_pk_bf16(pr[0], pr[1]);
                    pw[2 * it + 1] = cvt_pk_bf16(pr[2], pr[3]);
                    lp += (bf16lo_to_f32(pw[2 * it]) + bf16hi_to_f32(pw[2 * it])) + (bf16lo_to_f32(pw[2 * it + 1]) + bf16hi_to_f32(pw[2 * it + 1]));
                }
                pa[ip] = make_uint4(pw[0], pw[1], pw[2], pw[3]);
            }
            l_run = fmaf(l_run, alpha, lp);
            float av[4];
    #pragma unroll
            for (int v = 0; v < 4; ++v) av[v] = __int_as_float(__builtin_amdgcn_ds_bpermute(4 * (4 * kg + v), __float_as_int(alpha)));
    #pragma unroll
            for (int n2 = 0; n2 < 2; ++n2) {
                const int ch = 32 * wave + 16 * n2 + c;
                f32x4_t on = o[n2];
    #pragma unroll
                for (int v = 0; v < 4; ++v) on[v] *= av[v];
    #pragma unroll
                for (int ip = 0; ip < 2; ++ip) {
                    uint4 vv = *reinterpret_cast<const uint4*>(Kb + pal_v_read_off(ch, ip, kg));
                    const int tk = tb + 32 * ip + 8 * kg;
                    vv.x &= (tk + 0 < t1 ? 0x0000FFFFu : 0u) | (tk + 1 < t1 ? 0xFFFF0000u : 0u);
                    vv.y &= (tk + 2 < t1 ? 0x0000FFFFu : 0u) | (tk + 3 < t1 ? 0xFFFF0000u : 0u);
                    vv.z &= (tk + 4 < t1 ? 0x0000FFFFu : 0u) | (tk + 5 < t1 ? 0xFFFF0000u : 0u);
                    vv.w &= (tk + 6 < t1 ? 0x0000FFFFu : 0u) | (tk + 7 < t1 ? 0xFFFF0000u : 0u);
                    on = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, pa[ip]), __builtin_bit_cast(bf16x8_t, vv), on, 0, 0, 0);
                }
                o[n2] = on;
            }
            if (f + 1 >= pe || i + 1 >= ns) {
                // ---- last stage of this sequence in this share: the partial of (sequence b, workgroup w)
                float lt = l_run + __shfl_xor(l_run, 16, 64);
                lt += __shfl_xor(lt, 32, 64);
    #pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const int head = 4 * kg + v;
                    const float lh = __shfl(lt, head, 64), mh = __shfl(m_run, head, 64);
                    if (head < G) {
                        const int64_t pi = ((int64_t)b * p.H + hk * G + head) * p.max_partitions + w;
                        const float inv = lh > 0.f ? 1.f / lh : 0.f;
    #pragma unroll
                        for (int n2 = 0; n2 < 2; ++n2) p.tmp_out[pi * D + 32 * wave + 16 * n2 + c] = o[n2][v] * inv;
                        if (wave == 0 && c == 0) { p.max_logits[pi] = mh; p.exp_sums[pi] = lh; }
                    }
                }
            }
        }
    }
    f_lo = f_hi;
    }                                                                 // runs
    // (A last-arriver merge in this kernel instead of the reduce launch was measured in round 4 and lost: ragged batch 32 5800 tok/s against
    //  6177 with the reduce launch, profiles/r04_b32_stream_ab.txt -- the merge walks up to W slots per head behind the slowest share.)
}

// merge of the stream kernel's partials: workgroup (h, b) recomputes the cuts, finds the workgroups whose share met sequence b and
// combines their slots.  128 threads = the 128 channels.
__global__ void __launch_bounds__(128) paged_attn_stream_reduce_kernel(void* __restrict__ out, const float* __restrict__ tmp_out,
                                                                       const float* __restrict__ max_logits, const float* __restrict__ exp_sums,
                                                                       const uint32_t* __restrict__ context_lens, const int B, const int H,
                                                                       const int W, const int slots) {
    constexpr int D = 128;
    const int h = blockIdx.x, b = blockIdx.y;
    const int lane = threadIdx.x & 63, d = threadIdx.x;
    const int n_l = lane < B ? ((int)context_lens[lane] + 63) >> 6 : 0;
    int pend = n_l;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(pend, o, 64);
        if (lane >= o) pend += t;
    }
    const int S = __builtin_amdgcn_readlane(pend, 63);
    const int pe = __builtin_amdgcn_readlane(pend, b), ps = pe - __builtin_amdgcn_readlane(n_l, b);
    // lane = workgroup: does its share [cut(w), cut(w + 1)) meet [ps, pe)?
    bool meets = false;
    if (lane < W) {
        const int lo = (int)pas_cut(S, W, lane), hi = (int)pas_cut(S, W, lane + 1);
        meets = lo < hi && lo < pe && hi > ps;
    }
    uint64_t mask = __ballot(meets);
    const int64_t base = ((int64_t)b * H + h) * slots;
    float M = -1e30f;
    for (uint64_t mm = mask; mm; mm &= mm - 1) M = fmaxf(M, max_logits[base + __ffsll((unsigned long long)mm) - 1]);
    float den = 0.f, acc = 0.f;
    for (uint64_t mm = mask; mm; mm &= mm - 1) {
        const int wq = __ffsll((unsigned long long)mm) - 1;
        const float wt = exp_sums[base + wq] * __expf(max_logits[base + wq] - M);
        den += wt;
        acc = fmaf(tmp_out[(base + wq) * D + d], wt, acc);
    }
    if (mask) static_cast<uint16_t*>(out)[((int64_t)b * H + h) * D + d] = f32_to_bf16(den > 0.f ? acc / den : 0.f);
}

template <int D32, int WPB>
static int launch_mfma_w(const PAParams& p, int B, int P, hipStream_t st) {
    dim3 grid(p.Hkv, B, (P + WPB - 1) / WPB), block(64 * WPB);
    const int nt = p.partition_size / 16;
    const size_t shm = WPB > 1 ? (size_t)WPB * (p.H / p.Hkv) * (32 * D32 + 2) * sizeof(float) : 0;
    if (p.partition_size >= 256) {                                  // chunks of 32-token pairs, looped inside the wave (NT = 0)
        if constexpr (WPB == 1 || WPB == 4) {
            if (p.kv8) hipLaunchKernelGGL((paged_attn_mfma_kernel<D32, 0, true, WPB>), grid, block, shm, st, p);
            else hipLaunchKernelGGL((paged_attn_mfma_kernel<D32, 0, false, WPB>), grid, block, shm, st, p);
            return (int)hipGetLastError();
        } else {
            return (int)hipErrorInvalidValue;
        }
    }
    if (p.kv8) {
        if (nt == 2) hipLaunchKernelGGL((paged_attn_mfma_kernel<D32, 2, true, WPB>), grid, block, shm, st, p);
        else if (nt == 4) hipLaunchKernelGGL((paged_attn_mfma_kernel<D32, 4, true, WPB>), grid, block, shm, st, p);
        else if (nt == 8 && WPB == 1) hipLaunchKernelGGL((paged_attn_mfma_kernel<D32, 8, true, 1>), grid, block, shm, st, p);
        else return (int)hipErrorInvalidValue;
        return (int)hipGetLastError();
    }
    if (nt == 2) hipLaunchKernelGGL((paged_attn_mfma_kernel<D32, 2, false, WPB>), grid, block, shm, st, p);
    else if (nt == 4) hipLaunchKernelGGL((paged_attn_mfma_kernel<D32, 4, false, WPB>), grid, block, shm, st, p);
    else if (nt == 8 && WPB == 1) hipLaunchKernelGGL((paged_attn_mfma_kernel<D32, 8, false, 1>), grid, block, shm, st, p);
    else return (int)hipErrorInvalidValue;
    return (int)hipGetLastError();
}
template <int D32>
static int launch_mfma(const PAParams& p, int B, int P, int wpb, hipStream_t st) {
    if (wpb == 16) return launch_mfma_w<D32, 16>(p, B, P, st);
    if (wpb == 8) return launch_mfma_w<D32, 8>(p, B, P, st);
    return wpb == 4 ? launch_mfma_w<D32, 4>(p, B, P, st) : launch_mfma_w<D32, 1>(p, B, P, st);
}

// ------------------------------------------------------------------------------------------------ launchers
template <int LPT, int KVT, bool PART>
static int launch_flash_g(const PAParams& p, int B, int P, hipStream_t st) {
    const int G = p.H / p.Hkv;
    constexpr int NWV = PART ? 1 : 4;                   // v2: one wave per partition; v1: 4 waves + LDS merge
    constexpr int UNR = PART ? 8 : 4;
    dim3 grid(p.Hkv, B, P), block(64 * NWV);
    if (G <= 1) hipLaunchKernelGGL((paged_attn_flash_kernel<LPT, 1, KVT, PART, NWV, UNR>), grid, block, 0, st, p);
    else if (G <= 2) hipLaunchKernelGGL((paged_attn_flash_kernel<LPT, 2, KVT, PART, NWV, UNR>), grid, block, 0, st, p);
    else if (G <= 4) hipLaunchKernelGGL((paged_attn_flash_kernel<LPT, 4, KVT, PART, NWV, UNR>), grid, block, 0, st, p);
    else if (G <= 8) hipLaunchKernelGGL((paged_attn_flash_kernel<LPT, 8, KVT, PART, NWV, UNR>), grid, block, 0, st, p);
    else return (int)hipErrorInvalidValue;
    return (int)hipGetLastError();
}
template <int KVT, bool PART>
static int launch_flash(const PAParams& p, int B, int P, hipStream_t st) {
    if (p.D <= 64) return launch_flash_g<8, KVT, PART>(p, B, P, st);
    if (p.D <= 128) return launch_flash_g<16, KVT, PART>(p, B, P, st);
    if (p.D <= 256) return launch_flash_g<32, KVT, PART>(p, B, P, st);
    return (int)hipErrorInvalidValue;
}

#define PA_ARRIVE_SLOTS 65536
static int g_pa_fused = 1;                                          // mi355_set_tuning(3, 0) -> separate reduce launch
static int g_pa_wpb = 0;                                            // mi355_set_tuning(8, 1 | 4): waves (partitions) per workgroup, 0 = auto
// mi355_set_tuning(44, v) -- which kernel serves a partition size on the PAGED bf16 layout:
//   1 (default): 256 / 512 -> one wave walks its chunk (pa_mfma_chunk); 64 -> the balanced LDS-DMA stream when the launch has >= 64
//                (sequence, kv head) pairs (measured on the MI355X: ragged batch 32 +3.2 %, batch 1 -13 %: profiles/r04_b32_stream_ab.txt)
//   0: neither (256 / 512 go to the generic kernel, 64 to the one-partition waves);  5: chunks, never the stream (A/B)
//   3: the stream for every launch with partition size 64 (tests: small shapes);  2: + 1024 / 2048 / 4096 take the chunked LDS-DMA kernel
static int g_pa_loop = 1;
#define PA_STREAM_MIN_PAIRS 64
// internal (host layer): the next v2 call that takes the stream leaves its partials unmerged and reports its workgroup count, so that the
// merge can run fused with the staging of the next mat-mul's activation image (qmatmul.hip: mi355_internal_pa_stream_reduce_to_image)
static thread_local int g_pa_stream_noreduce = 0, g_pa_stream_last_w = 0;
static bool pa_stream_shape_ok(int B, int H, int Hkv, int D, int block_size) {
    return D == 128 && Hkv > 0 && H % Hkv == 0 && H / Hkv <= 16 && (block_size == 16 || block_size == 32 || block_size == 64) && B <= 64;
}
// what the step drivers ask: does a decode launch of this shape take the stream (-> they pass partition size 64)?
extern "C" int mi355_pa_stream_auto(int32_t num_seqs, int32_t num_heads, int32_t num_kv_heads, int32_t head_dim, int32_t block_size) {
    if (!(g_pa_loop == 1 || g_pa_loop == 2 || g_pa_loop == 3)) return 0;
    if (!pa_stream_shape_ok(num_seqs, num_heads, num_kv_heads, head_dim, block_size)) return 0;
    return (g_pa_loop == 3 || (int64_t)num_seqs * num_kv_heads >= PA_STREAM_MIN_PAIRS) ? 1 : 0;
}

static int pa_dispatch(PAParams p, int B, int P, int layout, int dtype, int64_t stream) {
    if (B <= 0) return 0;
    if (p.H % p.Hkv || (p.D & 7) || p.D > 256) return (int)hipErrorInvalidValue;
    if (dtype != MI355_DTYPE_BF16 && dtype != MI355_DTYPE_F16) return (int)hipErrorInvalidValue;
    hipStream_t st = to_stream(stream);
    int rc, wpb = 1;
    if (p.kv8 && (layout != MI355_KV_PAGED || dtype != MI355_DTYPE_BF16 || (p.D % 16))) return (int)hipErrorInvalidValue;
    if (layout == MI355_KV_FLASH) {
        if (P > 1)
            rc = (dtype == MI355_DTYPE_BF16) ? launch_flash<MI355_DTYPE_BF16, true>(p, B, P, st)
                                             : launch_flash<MI355_DTYPE_F16, true>(p, B, P, st);
        else
            rc = (dtype == MI355_DTYPE_BF16) ? launch_flash<MI355_DTYPE_BF16, false>(p, B, P, st)
                                             : launch_flash<MI355_DTYPE_F16, false>(p, B, P, st);
    } else if (layout == MI355_KV_PAGED && dtype == MI355_DTYPE_BF16 && p.kv8 && PAS_TS != 0 && p.partition_size == 64 && P > 1 &&
               mi355_pa_stream_auto(B, p.H, p.Hkv, p.D, p.block_size)) {
        // the e4m3fn cache through the same balanced stream (round 5): 16 KiB stages, a ring of 3; the merge scratch of > 8 query heads per
        // kv head has its own 32 KiB (a stage is too small for it).  A stage of this kernel lasts as long as a bf16 stage of twice the bytes
        // (1.8 us: the chain K read -> convert -> QK -> softmax shuffles -> P.V of one wave per SIMD, not the DMA: rings of 5 and 6 stages
        // changed nothing), so TWO workgroups share a CU where their LDS fits (64 KiB each at <= 8 query heads per kv head).  Measured on the
        // Mixtral-shaped batch-32 step, one box: 32-token MFMA partitions 8.03 ms, this stream one workgroup per CU 8.36, two 7.89.
        constexpr int PAS_R8 = 3;
        const int G = p.H / p.Hkv;
        // dynamic LDS; the kernel also holds 512 B of STATIC LDS (pas_stt): two workgroups per CU need 2 x (64 KiB + 512 B) of the 160 KiB
        const size_t shm8 = (size_t)PAS_R8 * 16384 + 16384 + (G > 8 ? 32768 : 0);
        static Mi355DevOnce attr_done;
        if (!attr_done.done()) {
            (void)hipFuncSetAttribute((const void*)paged_attn_stream_kernel<PAS_R8, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, PAS_R8 * 16384 + 16384 + 32768);
            attr_done.set();
        }
        int W = (G > 8 ? 256 : 512) / p.Hkv;
        if (W < 1) W = 1;
        if (W > 64) W = 64;
        if (W > P) W = P;
        hipLaunchKernelGGL((paged_attn_stream_kernel<PAS_R8, true, true>), dim3(W, p.Hkv), dim3(256), shm8, st, p, B, p.block_tables, p.context_lens);
        if (g_pa_stream_noreduce) { g_pa_stream_last_w = W; return (int)hipGetLastError(); }   // the caller merges
        hipLaunchKernelGGL(paged_attn_stream_reduce_kernel, dim3(p.H, B), dim3(128), 0, st, p.out, p.tmp_out, p.max_logits, p.exp_sums,
                           p.context_lens, B, p.H, W, p.max_partitions);
        return (int)hipGetLastError();
    } else if (layout == MI355_KV_PAGED && dtype == MI355_DTYPE_BF16 && !p.kv8 && p.partition_size == 64 && P > 1 &&
               mi355_pa_stream_auto(B, p.H, p.Hkv, p.D, p.block_size)) {
        // one balanced stream of 64-token stages per workgroup, W workgroups per kv head; partials merged by the reduce launch.
        // Measured on the ragged batch-32 step (round 4, each pair alternated on one box; profiles/r04_b32_stream_ab.txt):
        //   channel split, ring of 4, one workgroup per CU 6218 / 6238 tok/s  ->  ring of 2, two workgroups per CU 6361 / 6386
        //   token split (the waves share a stage's tokens instead of its output channels), ring of 2, two per CU: the same (6332 vs 6335)
        //   token split, ring of 3 / 4, one workgroup per CU: 6369 / 6360 against 6312 -- the default
        constexpr int PAS_R = PAS_RING;
        static Mi355DevOnce attr_done;
        if (!attr_done.done()) {
            (void)hipFuncSetAttribute((const void*)paged_attn_stream_kernel<PAS_R, PAS_TS != 0>, hipFuncAttributeMaxDynamicSharedMemorySize, PAS_R * 32768 + 16384);
            attr_done.set();
        }
        int W = (PAS_RING <= 2 ? 512 : 256) / p.Hkv;                  // two workgroups per CU fit with a ring of two stages only
        if (W < 1) W = 1;
        if (W > 64) W = 64;
        if (W > P) W = P;
        hipLaunchKernelGGL((paged_attn_stream_kernel<PAS_R, PAS_TS != 0>), dim3(W, p.Hkv), dim3(256), PAS_R * 32768 + 16384, st, p, B, p.block_tables,
                           p.context_lens);
        if (g_pa_stream_noreduce) { g_pa_stream_last_w = W; return (int)hipGetLastError(); }   // the caller merges (mi355_internal_pa_stream_partials)
        hipLaunchKernelGGL(paged_attn_stream_reduce_kernel, dim3(p.H, B), dim3(128), 0, st, p.out, p.tmp_out, p.max_logits, p.exp_sums,
                           p.context_lens, B, p.H, W, p.max_partitions);
        return (int)hipGetLastError();
    } else if (layout == MI355_KV_PAGED && dtype == MI355_DTYPE_BF16 && p.D == 128 && !p.kv8 && g_pa_loop == 2 && p.H / p.Hkv <= 16 &&
               (p.block_size == 16 || p.block_size == 32 || p.block_size == 64) &&
               (p.partition_size == 1024 || p.partition_size == 2048 || p.partition_size == 4096) && p.partition_size / p.block_size <= 64) {
        // EXPERIMENT (tuning key 44 = 2): chunks of 64-token stages through an LDS ring filled by DMA; partials merged by the reduce launch
        constexpr int PAL_R = 4;
        static Mi355DevOnce attr_done;
        if (!attr_done.done()) {
            (void)hipFuncSetAttribute((const void*)paged_attn_lds_kernel<PAL_R>, hipFuncAttributeMaxDynamicSharedMemorySize, PAL_R * 32768);
            attr_done.set();
        }
        hipLaunchKernelGGL((paged_attn_lds_kernel<PAL_R>), dim3(p.Hkv, B, P), dim3(256), PAL_R * 32768, st, p);
        rc = (int)hipGetLastError();
    } else if (layout == MI355_KV_PAGED && dtype == MI355_DTYPE_BF16 && (p.D == 128 || p.D == 64) &&
               p.H / p.Hkv <= 16 && (p.block_size % 16) == 0 &&
               (p.partition_size == 32 || p.partition_size == 64 || p.partition_size == 128 ||
                (g_pa_loop && (p.partition_size == 256 || p.partition_size == 512) && p.partition_size % p.block_size == 0 &&
                 p.partition_size / p.block_size <= 64))) {
        bool fused = false;
        const bool loop = p.partition_size >= 256;                  // pa_mfma_chunk: one wave walks its chunk pair by pair
        // few sequences, many partitions: 4 partitions per workgroup, merged in LDS (4 x fewer partials to reduce)
        // one sequence with a long context (batch-1 decode): 8 partitions per workgroup and the merge in the last arriver --
        // measured +1 % on the whole step in three sessions (507 -> 512 tok/s); at 2..8 sequences it is noise, so the rule stops there
        const bool lone = (int64_t)B * p.Hkv <= 8 && P >= 32 && p.partition_size <= 64 && g_pa_fused == 1 && g_pa_wpb == 0;
        if (g_pa_wpb > 0) wpb = ((g_pa_wpb == 4 || g_pa_wpb == 8 || g_pa_wpb == 16) && p.partition_size <= 64) ? g_pa_wpb : 1;
        else wpb = lone ? 8 : ((P >= 8 && p.partition_size <= 64) ? 4 : 1);
        if (loop) wpb = P > 1 ? 4 : 1;                              // four chunks per workgroup, merged in LDS
        // launch_mfma_w needs WPB * G * (32 * D32 + 2) * 4 bytes of dynamic LDS and no larger-LDS attribute is set for it: with 16 query
        // heads per kv head and D = 128 eight partitions per workgroup would ask for 66 560 B and the launch fails
        {
            const int G = p.H / p.Hkv, D32 = p.D / 32;
            while (wpb > 1 && (size_t)wpb * G * (32 * D32 + 2) * 4 > 64 * 1024) wpb = wpb == 16 ? 8 : (wpb == 8 ? 4 : 1);
        }
        // the in-kernel merge is one wave per (sequence, kv head): worth it only when there are many of them
        if (P > 1 && g_pa_fused && ((int64_t)B * p.Hkv >= (g_pa_fused > 1 ? 1 : 64) || lone) && (int64_t)B * p.Hkv <= PA_ARRIVE_SLOTS) {
            // tickets belong to (device, stream): two streams never share a counter (scratch.cpp)
            void* arr = nullptr;
            const int arc = mi355_scratch_get(&arr, MI355_SCR_PA_ARRIVE, PA_ARRIVE_SLOTS * 4, st, true);
            if (arc) return arc;
            p.arrive = static_cast<unsigned*>(arr);
            fused = true;
        }
        rc = (p.D == 128) ? launch_mfma<4>(p, B, P, wpb, st) : launch_mfma<2>(p, B, P, wpb, st);
        if (fused) return rc;
    } else if (layout == MI355_KV_PAGED) {
        const size_t shm = (size_t)p.partition_size * sizeof(float);
        if (shm > 60 * 1024) return (int)hipErrorInvalidValue;   // caller must partition (v2) long contexts
        dim3 grid(p.H, B, P), block(256);
        if (dtype == MI355_DTYPE_BF16)
            hipLaunchKernelGGL(paged_attn_paged_layout_kernel<MI355_DTYPE_BF16>, grid, block, shm, st, p);
        else
            hipLaunchKernelGGL(paged_attn_paged_layout_kernel<MI355_DTYPE_F16>, grid, block, shm, st, p);
        rc = (int)hipGetLastError();
    } else {
        return (int)hipErrorInvalidValue;
    }
    if (rc != 0 || P <= 1) return rc;
    dim3 rgrid(p.H, B), rblock(512);
    const size_t rshm = (size_t)(p.max_partitions > 512 ? p.max_partitions : 512) * sizeof(float);
    if (rshm > 60 * 1024) return (int)hipErrorInvalidValue;
    if (dtype == MI355_DTYPE_BF16)
        hipLaunchKernelGGL(paged_attn_reduce_kernel<MI355_DTYPE_BF16>, rgrid, rblock, rshm, st, p.out, p.tmp_out,
                           p.max_logits, p.exp_sums, p.context_lens, p.H, p.D, p.partition_size * wpb, p.max_partitions);
    else
        hipLaunchKernelGGL(paged_attn_reduce_kernel<MI355_DTYPE_F16>, rgrid, rblock, rshm, st, p.out, p.tmp_out,
                           p.max_logits, p.exp_sums, p.context_lens, p.H, p.D, p.partition_size * wpb, p.max_partitions);
    return (int)hipGetLastError();
}

// host-side view of the LDS stage layout of paged_attn_lds_kernel (tests only): what = 0 row (channel group / channel) and 1 first token
// of the 16 bytes DMA piece a, lane b copies; 2 byte offset of the K fragment read (j, kg, ip, it, r) = (a, b, c, d, e); 3 of the V
// fragment read (channel, ip, kg) = (a, b, c); 4 the stream kernel's cut (S, W, w) = (a, b, c); 5..8 the same four views of the e4m3fn stage
extern "C" int32_t mi355_internal_pal_layout(int32_t what, int32_t a, int32_t b, int32_t c, int32_t d, int32_t e) {
    switch (what) {
    case 0: return pal_dma_row(a, b);
    case 1: return pal_dma_token(a, b);
    case 2: return pal_k_read_off(a, b, c, d, e);
    case 3: return pal_v_read_off(a, b, c);
    case 4: return (int32_t)pas_cut(a, b, c);                         // first flat stage of workgroup c of b, S = a stages in all
    case 5: return pal8_dma_row(a, b);                                // the e4m3fn stage (16 pieces): row / first token of piece a, lane b
    case 6: return pal8_dma_token(a, b);
    case 7: return pal8_k_read_off(a, b, c, d, e);                    // 8-byte K fragment read (j, kg, ip, it, r)
    case 8: return pal8_v_read_off(a, b, c, d);                       // 4-byte V fragment read (channel, ip, kg, it)
    default: return -1;
    }
}


// ------------------------------------------------------------------------------------------------
// PARITY MODE (tests only; mi355_llama_set_attention_numerics(model, 1)): decode attention with the reference CPU path's own rounding
// points -- `NaiveAttention::forward` on bf16 tensors (models/mod.rs:1288-1306): the score matmul returns bf16, `* scale` rounds to
// bf16 again, softmax_last_dim returns bf16 probabilities, the P.V matmul accumulates in f32 and returns bf16 -- and the SAME
// summation orders as oracle/oracle.c's bf16-attention mode (sequential over d for a score, over t for the denominator in f64 and for
// every output channel), so that the two agree except where an f32 rounding difference happens to straddle a bf16 tie.  The product
// kernels keep scores / probabilities in f32 (more accurate than the reference); through 32 layers the reference's bf16 points alone
// move the logits by 1.7-3 %, so north_star's "within 1e-3 of the reference CPU logits" can only be checked in
// this mode.  One workgroup per (head, sequence); slow by design (a thread walks whole rows).
__global__ void __launch_bounds__(256) paged_attn_refnum_kernel(const PAParams p, const int flash) {
#pragma clang fp contract(fast)
    extern __shared__ float rn_sm[];                                  // [D] q | [n] scores -> probabilities
    const int h = blockIdx.x, b = blockIdx.y, D = p.D, G = p.H / p.Hkv, hk = h / G, bs = p.block_size;
    const int n = (int)p.context_lens[b];
    float* qs = rn_sm;
    float* sc = rn_sm + D;
    __shared__ float red[256];
    __shared__ double den_s;
    const uint16_t* q = static_cast<const uint16_t*>(p.q) + (size_t)b * p.q_stride + (size_t)h * D;
    for (int d = threadIdx.x; d < D; d += blockDim.x) qs[d] = bf16_to_f32(q[d]);
    __syncthreads();
    const uint16_t* kc = static_cast<const uint16_t*>(p.kc);
    const uint16_t* vc = static_cast<const uint16_t*>(p.vc);
    float mx = -1e30f;
    for (int t = threadIdx.x; t < n; t += blockDim.x) {
        const size_t blk = p.block_tables[(size_t)b * p.max_blocks + t / bs];
        const int off = t % bs;
        float s = 0.f;
        for (int d = 0; d < D; ++d) {
            const size_t ki = flash ? ((blk * bs + off) * p.Hkv + hk) * D + d
                                    : ((((blk * p.Hkv + hk) * (D / 8) + d / 8) * bs + off) * 8) + d % 8;
            s = fmaf(qs[d], bf16_to_f32(kc[ki]), s);
        }
        const float v = bf16_to_f32(f32_to_bf16(bf16_to_f32(f32_to_bf16(s)) * p.scale));
        sc[t] = v;
        mx = fmaxf(mx, v);
    }
    red[threadIdx.x] = mx;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]); __syncthreads(); }
    mx = red[0];
    for (int t = threadIdx.x; t < n; t += blockDim.x) sc[t] = expf(sc[t] - mx);
    __syncthreads();
    if (threadIdx.x == 0) {                                           // the oracle's order: sequential, f64
        double den = 0.0;
        for (int t = 0; t < n; ++t) den += (double)sc[t];
        den_s = den;
    }
    __syncthreads();
    const double den = den_s;
    for (int t = threadIdx.x; t < n; t += blockDim.x) sc[t] = bf16_to_f32(f32_to_bf16((float)((double)sc[t] / den)));
    __syncthreads();
    uint16_t* out = static_cast<uint16_t*>(p.out) + ((size_t)b * p.H + h) * D;
    for (int d = threadIdx.x; d < D; d += blockDim.x) {
        float o = 0.f;
        for (int t = 0; t < n; ++t) {
            const size_t blk = p.block_tables[(size_t)b * p.max_blocks + t / bs];
            const int off = t % bs;
            const size_t vi = flash ? ((blk * bs + off) * p.Hkv + hk) * D + d : ((blk * p.Hkv + hk) * D + d) * (size_t)bs + off;
            o = fmaf(sc[t], bf16_to_f32(vc[vi]), o);
        }
        out[d] = f32_to_bf16(o);
    }
}
extern "C" int mi355_paged_attention_reference_numerics(void* out, const void* q, const void* key_cache, const void* value_cache,
                                                        const uint32_t* block_tables, const uint32_t* context_lens, int32_t num_seqs,
                                                        int32_t num_heads, int32_t num_kv_heads, int32_t head_dim, int32_t block_size,
                                                        int32_t max_blocks_per_seq, int32_t max_context_len, float scale, int32_t layout,
                                                        int64_t stream) {
    if (num_seqs <= 0) return 0;
    if (!out || !q || !key_cache || !value_cache || !block_tables || !context_lens || num_kv_heads <= 0 || num_heads % num_kv_heads ||
        (head_dim & 7) || head_dim > 256 || (layout != MI355_KV_PAGED && layout != MI355_KV_FLASH))
        return (int)hipErrorInvalidValue;
    const size_t shm = ((size_t)head_dim + (size_t)(max_context_len > 0 ? max_context_len : 1)) * sizeof(float);
    if (shm > 150 * 1024) return (int)hipErrorInvalidValue;
    static Mi355DevOnce attr_done;
    if (!attr_done.done()) {
        (void)hipFuncSetAttribute((const void*)paged_attn_refnum_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        attr_done.set();
    }
    PAParams p{};
    p.out = out; p.q = q; p.kc = key_cache; p.vc = value_cache; p.block_tables = block_tables; p.context_lens = context_lens;
    p.H = num_heads; p.Hkv = num_kv_heads; p.D = head_dim; p.block_size = block_size; p.max_blocks = max_blocks_per_seq;
    p.scale = scale; p.q_stride = (int64_t)num_heads * head_dim;
    hipLaunchKernelGGL(paged_attn_refnum_kernel, dim3(num_heads, num_seqs), dim3(256), shm, to_stream(stream), p, layout == MI355_KV_FLASH ? 1 : 0);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// `sliding_window` of PagedAttention::new on decode steps (attention.rs:566-575,888-897; kernel in the un-vendored attention-rs): the
// query (position ctx - 1) sees the last `window` keys only, positions ctx - window .. ctx - 1 -- what vLLM's paged attention gets by
// truncating the block table to the window, here exact for windows that do not start on a block boundary.  No BASELINE model carries
// a window: this is the correctness path (one workgroup per (head, sequence), the window's scores in LDS, f32 softmax as the product
// kernels), every head size <= 256, both cache layouts, bf16 / f16.
template <int KVT>
__global__ void __launch_bounds__(256) paged_attn_window_kernel(const PAParams p, const int window, const int flash) {
    extern __shared__ float pw_sm[];                                  // [D] q * scale | [<= window] scores -> probabilities
    __shared__ float red[16];
    const int h = blockIdx.x, b = blockIdx.y, D = p.D, G = p.H / p.Hkv, hk = h / G, bs = p.block_size;
    const int ctx = (int)p.context_lens[b];
    uint16_t* out = static_cast<uint16_t*>(p.out) + ((size_t)b * p.H + h) * D;
    if (ctx <= 0) { for (int d = threadIdx.x; d < D; d += blockDim.x) out[d] = 0; return; }
    const int t_lo = ctx > window ? ctx - window : 0, n = ctx - t_lo;
    float* qs = pw_sm;
    float* sc = pw_sm + D;
    auto cvt = [](uint16_t v) { return KVT == MI355_DTYPE_BF16 ? bf16_to_f32(v) : f16_bits_to_f32(v); };
    const uint16_t* q = static_cast<const uint16_t*>(p.q) + (size_t)b * p.q_stride + (size_t)h * D;
    for (int d = threadIdx.x; d < D; d += blockDim.x) qs[d] = cvt(q[d]) * p.scale;
    __syncthreads();
    const uint16_t* kc = static_cast<const uint16_t*>(p.kc);
    const uint16_t* vc = static_cast<const uint16_t*>(p.vc);
    const uint32_t* bt = p.block_tables + (size_t)b * p.max_blocks;
    float mx = -1e30f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const int t = t_lo + i;
        const size_t blk = bt[t / bs];
        const int off = t % bs;
        float s = 0.f;
        for (int d = 0; d < D; ++d) {
            const size_t ki = flash ? ((blk * bs + off) * p.Hkv + hk) * D + d
                                    : ((((blk * p.Hkv + hk) * (D / 8) + d / 8) * bs + off) * 8) + d % 8;
            s = fmaf(qs[d], cvt(kc[ki]), s);
        }
        if (p.softcap > 0.f) s = tanhf(s / p.softcap) * p.softcap;
        sc[i] = s;
        mx = fmaxf(mx, s);
    }
    {
        const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
        mx = wave_max(mx);
        if (lane == 0) red[wid] = mx;
        __syncthreads();
        mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        __syncthreads();
    }
    float lsum = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float e = __expf(sc[i] - mx);
        sc[i] = e;
        lsum += e;
    }
    lsum = block_sum(lsum, red);                                      // (its barrier publishes the probabilities)
    for (int d = threadIdx.x; d < D; d += blockDim.x) {
        float a = 0.f;
        for (int i = 0; i < n; ++i) {
            const int t = t_lo + i;
            const size_t blk = bt[t / bs];
            const int off = t % bs;
            const size_t vi = flash ? ((blk * bs + off) * p.Hkv + hk) * D + d : ((blk * p.Hkv + hk) * D + d) * (size_t)bs + off;
            a = fmaf(sc[i], cvt(vc[vi]), a);
        }
        const float o = a / lsum;
        out[d] = (KVT == MI355_DTYPE_BF16) ? f32_to_bf16(o) : f32_to_f16_bits(o);
    }
}
extern "C" int mi355_paged_attention_window(void* out, const void* q, const void* key_cache, const void* value_cache,
                                            const uint32_t* block_tables, const uint32_t* context_lens, int32_t num_seqs,
                                            int32_t num_heads, int32_t num_kv_heads, int32_t head_dim, int32_t block_size,
                                            int32_t max_blocks_per_seq, int32_t max_context_len, float scale, float softcap,
                                            int32_t layout, int32_t dtype, int32_t sliding_window, int64_t stream) {
    if (sliding_window <= 0)
        return mi355_paged_attention_v1(out, q, key_cache, value_cache, block_tables, context_lens, num_seqs, num_heads, num_kv_heads, head_dim,
                                        block_size, max_blocks_per_seq, max_context_len, scale, softcap, layout, dtype, stream);
    if (num_seqs <= 0) return 0;
    if (!out || !q || !key_cache || !value_cache || !block_tables || !context_lens || num_kv_heads <= 0 || num_heads % num_kv_heads ||
        head_dim <= 0 || (head_dim & 7) || head_dim > 256 || block_size <= 0 || (layout != MI355_KV_PAGED && layout != MI355_KV_FLASH) ||
        (dtype != MI355_DTYPE_BF16 && dtype != MI355_DTYPE_F16))
        return (int)hipErrorInvalidValue;
    const int span = sliding_window < max_context_len ? sliding_window : (max_context_len > 0 ? max_context_len : 1);
    const size_t shm = ((size_t)head_dim + (size_t)span) * sizeof(float);
    if (shm > 150 * 1024) return (int)hipErrorInvalidValue;           // windows beyond ~38 k tokens: not served by this path
    static Mi355DevOnce attr_done;
    if (!attr_done.done()) {
        (void)hipFuncSetAttribute((const void*)paged_attn_window_kernel<MI355_DTYPE_BF16>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        (void)hipFuncSetAttribute((const void*)paged_attn_window_kernel<MI355_DTYPE_F16>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        attr_done.set();
    }
    PAParams p{};
    p.out = out; p.q = q; p.kc = key_cache; p.vc = value_cache; p.block_tables = block_tables; p.context_lens = context_lens;
    p.H = num_heads; p.Hkv = num_kv_heads; p.D = head_dim; p.block_size = block_size; p.max_blocks = max_blocks_per_seq;
    p.scale = scale; p.softcap = softcap; p.q_stride = (int64_t)num_heads * head_dim;
    const int flash = layout == MI355_KV_FLASH ? 1 : 0;
    if (dtype == MI355_DTYPE_BF16)
        hipLaunchKernelGGL(paged_attn_window_kernel<MI355_DTYPE_BF16>, dim3(num_heads, num_seqs), dim3(256), shm, to_stream(stream), p, sliding_window, flash);
    else
        hipLaunchKernelGGL(paged_attn_window_kernel<MI355_DTYPE_F16>, dim3(num_heads, num_seqs), dim3(256), shm, to_stream(stream), p, sliding_window, flash);
    return (int)hipGetLastError();
}

/* decode attention through the balanced stream WITHOUT its merge launch: partials stay in tmp_out / max_logits / exp_sums (slot w of
 * max_partitions per (sequence, head)); *w_out = workgroups per kv head (0: the launch did not take the stream and is complete). */
extern "C" int mi355_internal_paged_attention_v2_partials(void* out, float* exp_sums, float* max_logits, float* tmp_out, const void* q,
                                                          const void* key_cache, const void* value_cache, const uint32_t* block_tables,
                                                          const uint32_t* context_lens, int32_t num_seqs, int32_t num_heads,
                                                          int32_t num_kv_heads, int32_t head_dim, int32_t block_size,
                                                          int32_t max_blocks_per_seq, int32_t max_context_len, int32_t partition_size,
                                                          float scale, float softcap, int32_t layout, int32_t dtype, int64_t stream,
                                                          int32_t* w_out) {
    g_pa_stream_noreduce = 1; g_pa_stream_last_w = 0;
    const int rc = mi355_paged_attention_v2(out, exp_sums, max_logits, tmp_out, q, key_cache, value_cache, block_tables, context_lens, num_seqs,
                                            num_heads, num_kv_heads, head_dim, block_size, max_blocks_per_seq, max_context_len, partition_size,
                                            scale, softcap, layout, dtype, stream);
    g_pa_stream_noreduce = 0;
    if (w_out) *w_out = g_pa_stream_last_w;
    return rc;
}

/* the same over the e4m3fn cache */
extern "C" int mi355_internal_paged_attention_fp8_partials(void* out, float* exp_sums, float* max_logits, float* tmp_out, const void* q,
                                                           const void* key_cache, const void* value_cache, const uint32_t* block_tables,
                                                           const uint32_t* context_lens, int32_t num_seqs, int32_t num_heads,
                                                           int32_t num_kv_heads, int32_t head_dim, int32_t block_size,
                                                           int32_t max_blocks_per_seq, int32_t max_context_len, int32_t partition_size,
                                                           float scale, float softcap, float k_scale, float v_scale, int64_t stream, int32_t* w_out) {
    g_pa_stream_noreduce = 1; g_pa_stream_last_w = 0;
    const int rc = mi355_paged_attention_fp8(out, exp_sums, max_logits, tmp_out, q, key_cache, value_cache, block_tables, context_lens, num_seqs,
                                             num_heads, num_kv_heads, head_dim, block_size, max_blocks_per_seq, max_context_len, partition_size,
                                             scale, softcap, k_scale, v_scale, stream);
    g_pa_stream_noreduce = 0;
    if (w_out) *w_out = g_pa_stream_last_w;
    return rc;
}

extern "C" int mi355_internal_pa_stream_reduce(void* out, const float* tmp_out, const float* max_logits, const float* exp_sums,
                                               const uint32_t* context_lens, int32_t B, int32_t H, int32_t W, int32_t slots, int64_t stream) {
    hipLaunchKernelGGL(paged_attn_stream_reduce_kernel, dim3(H, B), dim3(128), 0, to_stream(stream), out, tmp_out, max_logits, exp_sums,
                       context_lens, B, H, W, slots);
    return (int)hipGetLastError();
}

void mi355_pa_set_fused(int v) { g_pa_fused = v; }
void mi355_pa_set_wpb(int v) { g_pa_wpb = v; }
void mi355_pa_set_loop(int v) { g_pa_loop = v; }

// decode attention over an fp8 (e4m3fn) KV cache in the PAGED layout (K x = 16); partition_size 0 = one pass (v1)
extern "C" int mi355_paged_attention_fp8(void* out, float* exp_sums, float* max_logits, float* tmp_out, const void* q,
                                         const void* key_cache, const void* value_cache, const uint32_t* block_tables,
                                         const uint32_t* context_lens, int32_t num_seqs, int32_t num_heads,
                                         int32_t num_kv_heads, int32_t head_dim, int32_t block_size,
                                         int32_t max_blocks_per_seq, int32_t max_context_len, int32_t partition_size,
                                         float scale, float softcap, float k_scale, float v_scale, int64_t stream) {
    PAParams p{};
    p.out = out; p.tmp_out = tmp_out; p.max_logits = max_logits; p.exp_sums = exp_sums;
    p.q = q; p.kc = key_cache; p.vc = value_cache;
    p.block_tables = block_tables; p.context_lens = context_lens;
    p.H = num_heads; p.Hkv = num_kv_heads; p.D = head_dim; p.block_size = block_size; p.max_blocks = max_blocks_per_seq;
    p.kv8 = 1; p.k_scale = k_scale; p.v_scale = v_scale;
    p.scale = scale; p.softcap = softcap; p.q_stride = (int64_t)num_heads * head_dim;
    if (partition_size <= 0) {
        p.partition_size = max_context_len > 0 ? max_context_len : 1; p.max_partitions = 1;
        return pa_dispatch(p, num_seqs, 1, MI355_KV_PAGED, MI355_DTYPE_BF16, stream);
    }
    if (!exp_sums || !max_logits || !tmp_out) return (int)hipErrorInvalidValue;
    p.partition_size = partition_size;
    p.max_partitions = (max_context_len + partition_size - 1) / partition_size;
    if (p.max_partitions < 1) p.max_partitions = 1;
    return pa_dispatch(p, num_seqs, p.max_partitions, MI355_KV_PAGED, MI355_DTYPE_BF16, stream);
}

extern "C" int mi355_paged_attention_v1(void* out, const void* q, const void* key_cache, const void* value_cache,
                                        const uint32_t* block_tables, const uint32_t* context_lens,
                                        int32_t num_seqs, int32_t num_heads, int32_t num_kv_heads,
                                        int32_t head_dim, int32_t block_size, int32_t max_blocks_per_seq,
                                        int32_t max_context_len, float scale, float softcap, int32_t layout,
                                        int32_t dtype, int64_t stream) {
    PAParams p{};
    p.out = out; p.q = q; p.kc = key_cache; p.vc = value_cache;
    p.block_tables = block_tables; p.context_lens = context_lens;
    p.H = num_heads; p.Hkv = num_kv_heads; p.D = head_dim; p.block_size = block_size;
    p.max_blocks = max_blocks_per_seq;
    p.partition_size = max_context_len > 0 ? max_context_len : 1;
    p.max_partitions = 1;
    p.scale = scale; p.softcap = softcap; p.q_stride = (int64_t)num_heads * head_dim;
    return pa_dispatch(p, num_seqs, 1, layout, dtype, stream);
}

extern "C" int mi355_paged_attention_v2(void* out, float* exp_sums, float* max_logits, float* tmp_out,
                                        const void* q, const void* key_cache, const void* value_cache,
                                        const uint32_t* block_tables, const uint32_t* context_lens,
                                        int32_t num_seqs, int32_t num_heads, int32_t num_kv_heads,
                                        int32_t head_dim, int32_t block_size, int32_t max_blocks_per_seq,
                                        int32_t max_context_len, int32_t partition_size, float scale,
                                        float softcap, int32_t layout, int32_t dtype, int64_t stream) {
    if (partition_size <= 0) return (int)hipErrorInvalidValue;
    PAParams p{};
    p.out = out; p.tmp_out = tmp_out; p.max_logits = max_logits; p.exp_sums = exp_sums;
    p.q = q; p.kc = key_cache; p.vc = value_cache;
    p.block_tables = block_tables; p.context_lens = context_lens;
    p.H = num_heads; p.Hkv = num_kv_heads; p.D = head_dim; p.block_size = block_size;
    p.max_blocks = max_blocks_per_seq;
    p.partition_size = partition_size;
    p.max_partitions = (max_context_len + partition_size - 1) / partition_size;
    if (p.max_partitions < 1) p.max_partitions = 1;
    p.scale = scale; p.softcap = softcap; p.q_stride = (int64_t)num_heads * head_dim;
    return pa_dispatch(p, num_seqs, p.max_partitions, layout, dtype, stream);
}
