// Skinny (decode-shaped) linear layers on the reference's NATIVE weight layouts, for the safetensors path:
//   K10  dense 16-bit   `Linear::forward`  y = x.W^T (+b)        src/openai/models/linear.rs:124-172
//   K12  marlin_4bit_{bf16,f16}            y = x.(s*(q-8))       src/backend/gptq.rs:132-178
//   K13  marlin_awq_4bit_*                 y = x.(s*(q-z))       src/backend/gptq.rs:116-131,148-162
//   K14  gemm_half_q_half_alt (f16, act-order g_idx)             src/backend/gptq.rs:181-197
//   K15  gptq_repack / awq_repack                                src/backend/gptq.rs:313-332
// Decode is an HBM stream of the weights (2 B resp. ~0.53 B per weight, each read once); MFMA does the
// contraction with the tokens on the M rows (16 tokens per M tile; activations are exactly 16-bit so no hi/lo
// split is needed).  No LDS staging: weight fragments come straight from global memory in fragment order
// (a 16x16x32 B-fragment is 8 consecutive k of one output row = 16 B of a dense row, or ONE u32 of a GPTQ
// k-packed column), activation fragments straight from L2.
//
// Our packed 4-bit layout IS the GPTQ checkpoint layout qweight[k/8][n] (bits 4i.. = code of k = 8*row+i):
// `gptq_repack` is therefore a copy and `awq_repack` is the only real transform (AWQ packs along n with the
// interleave [0,2,4,6,1,3,5,7]).  Scales (and AWQ zero points) arrive Marlin-permuted from the reference's host
// code (`marlin_permute_scales`, linear.rs:341-379; examples/convert_awq_marlin.py:72-115) and are un-permuted
// by index arithmetic here.
#include "common.h"
#include "scratch.h"
#include "../../include/mi355_vllm.h"
#include <hip/hip_runtime.h>
#include <utility>

typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned int dg_u32x4 __attribute__((ext_vector_type(4)));

// DW_GPTQ4: checkpoint layout qweight[k/8][n] u32.  DW_GPTQ4T: the same words in 16-column x 256-k tiles (gptq_tile_index below):
// one wave's share of a k-block is two contiguous 1 KiB runs (two dwordx4 loads per lane) instead of eight loads that each
// touch four 64-byte segments a row apart -- what gptq_repack / awq_repack / mi355_marlin_format_repack produce at load time.
enum { DW_DENSE = 0, DW_GPTQ4 = 1, DW_GPTQ4T = 2 };
// word index of qweight[kr][n] (kr = k/8) in the tiled image: [tile n/16][k-block kr/32][half][lane = 16*(kr%4) + n%16][4]
// holding rows kr = 32 kb + 4 (4 half + jj) + lane/16 -- the fragment order of dense_body's 4-bit arm
__host__ __device__ __forceinline__ size_t gptq_tile_index(int kr, int n, int nkb) {
    const int kb = kr >> 5, rem = kr & 31, j = rem >> 2, kg = rem & 3;
    return ((((size_t)(n >> 4) * nkb + kb) * 2 + (j >> 2)) * 64 + kg * 16 + (n & 15)) * 4 + (j & 3);
}
// 16-bit weights [N][K] in the same 16-row x 256-k tiles (element index): [tile n/16][k-block k/256][j = (k%256)/32][lane = 16*((k%32)/8) + n%16][k%8]
// -- fragment j of a (tile, k-block) is 1 KiB contiguous, the whole (tile, k-block) 8 KiB; a row-major row tile is sixteen 512-byte
// pieces 2 K bytes apart, and a wave instruction touches 16 rows x 64 B of it (measured: 3.9 TB/s at best from that pattern)
__host__ __device__ __forceinline__ size_t dense_tile_index(int n, int k, int nkb) {
    const int kb = k >> 8, kk = k & 255, j = kk >> 5, kg = (kk & 31) >> 3, e = kk & 7;
    return (((((size_t)(n >> 4) * nkb + kb) * 8 + j) * 64 + kg * 16 + (n & 15)) << 3) + e;
}
enum { SP_NONE = 0, SP_GROUPED = 1, SP_SINGLE = 2 };

// attention.rs:644-719 after the projections: q,k -> f32 -> rope -> model dtype, then the cache write of k and v
struct DenseRope {
    const float* cos_t; const float* sin_t;         // [max_seq][head_dim / 2] f32
    const int64_t* positions; const int64_t* slots; // [T]
    uint16_t* kcache; uint16_t* vcache;             // this layer's bf16 cache
    int32_t n_kv_heads, head_dim, block_size, flash;
};
struct DenseArgs {
    const void* w;            // DENSE: 16-bit [N][ldw] (wtiled: the 16 x 256 tile image, dense_tile_index) ; GPTQ4: u32 [K/8][N]
    int32_t ldw;
    int32_t wtiled;           // DENSE only: 1 = w is the tiled image (N % 16 == 0, K % 256 == 0; ldw unused)
    const void* scales;       // GPTQ4: 16-bit [K/g][N]
    const uint32_t* qzeros;   // packed zero points [K/g][N/8] or null
    int32_t zmode, sperm;     // MI355_ZERO_* ; SP_*
    int32_t group_size;       // resolved: multiple of 32, <= K
    const void* x;            // 16-bit [T][ldx]
    int32_t ldx, T, N, K;
    void* out;                // 16-bit [T][ldo]
    int32_t ldo;
    const void* bias;         // 16-bit [N] or null
    const void* resid;        // 16-bit [T][ldo] (EPI_RESID)
    int32_t epi;              // MI355_EPI_STORE / RESID / SILU_MUL
    int32_t pair_offset;      // SILU_MUL: rows of `up` start at pair_offset (packed gate_up weight, mlp.rs:324-352)
    const void* norm_w;       // dense_small_kernel only: RmsNorm weight [K] (bf16) applied to x while it is staged, or null
    float norm_eps;
    const float* ss_in;       // with norm_w: [T][K/16] partial sums of squares of x's rows, left by the launch that produced x
    // dense_small3r_kernel only (q / k / v of a decode step): RoPE and the KV-cache write in the projection's epilogue
    int32_t rope_mode;        // 0 none; 1 = q (rotate, store); 2 = k (rotate, store, write the key cache); 3 = v (store, write the value cache)
    float* ss_out;            // dense_small_kernel only: where this launch leaves [T][ldo/16] partial sums of squares of its output
};

template <int DT> __device__ __forceinline__ float h2f(uint16_t h) {
    return DT == MI355_DTYPE_BF16 ? bf16_to_f32(h) : f16_bits_to_f32(h);
}
template <int DT> __device__ __forceinline__ uint16_t f2h(float f) {
    return DT == MI355_DTYPE_BF16 ? f32_to_bf16(f) : f32_to_f16_bits(f);
}
template <int DT> __device__ __forceinline__ float rnd(float f) { return h2f<DT>(f2h<DT>(f)); }

template <int DT>
__device__ __forceinline__ f32x4_t mfma32(const uint4& a, const uint4& b, const f32x4_t& c) {
    if constexpr (DT == MI355_DTYPE_BF16)
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
}

// position of natural column n inside a Marlin-permuted row (inverse of linear.rs:341-379)
//   grouped: new[64c + 8i + j] = old[64c + i + 8j]
//   single : new[32c + 8i + jj] = old[32c + 2i + {0,1,8,9,16,17,24,25}[jj]]
__host__ __device__ __forceinline__ int marlin_scale_pos(int n, int sperm) {
    if (sperm == SP_GROUPED) {
        const int c = n & 63;
        return (n & ~63) + 8 * (c & 7) + (c >> 3);
    }
    if (sperm == SP_SINGLE) {
        const int c = n & 31;
        return (n & ~31) + 8 * ((c >> 1) & 3) + 2 * (c >> 3) + (c & 1);
    }
    return n;
}

__device__ __forceinline__ float zero_point(const DenseArgs& a, int g, int col) {
    if (a.zmode == MI355_ZERO_SYM8 || !a.qzeros) return 8.f;
    if (a.zmode == MI355_ZERO_GPTQ_PLUS1)   // AutoGPTQ v1: stored zero = z - 1, nibble n%8 of qzeros[g][n/8]
        return (float)(((a.qzeros[(size_t)g * (a.N / 8) + col / 8] >> (4 * (col & 7))) & 0xF) + 1);
    // marlin zero points (convert_awq_marlin.py:72-91): scale_perm, then interleave [0,2,4,6,1,3,5,7], then pack
    const int p1 = marlin_scale_pos(col, SP_GROUPED);
    const int u = p1 & 7;
    const int t = ((u & 1) << 2) | (u >> 1);            // argsort([0,2,4,6,1,3,5,7]) = [0,4,1,5,2,6,3,7]
    const int p2 = (p1 & ~7) + t;
    return (float)((a.qzeros[(size_t)g * (a.N / 8) + p2 / 8] >> (4 * (p2 & 7))) & 0xF);
}

// the rounding chain of one output element: y (and the `up` value u of a gate/up pair) -> the model dtype
template <int DT>
__device__ __forceinline__ float dense_epilogue(const DenseArgs& a, const int m, const int row, const float y, const float up) {
    const uint16_t* b16 = static_cast<const uint16_t*>(a.bias);
    // candle rounds the matmul result to the model dtype, then the bias add rounds again (linear.rs:124-172)
    float o = rnd<DT>(y);
    if (b16) o = rnd<DT>(o + h2f<DT>(b16[row]));
    if (a.epi == MI355_EPI_SILU_MUL) {
        float u = rnd<DT>(up);
        if (b16) u = rnd<DT>(u + h2f<DT>(b16[a.pair_offset + row]));
        o = rnd<DT>(rnd<DT>(o / (1.f + __expf(-o))) * u);               // silu(gate) * up (mlp.rs:457)
    } else if (a.epi == MI355_EPI_RESID) {
        o = rnd<DT>(o + h2f<DT>(static_cast<const uint16_t*>(a.resid)[(size_t)m * a.ldo + row]));
    }
    static_cast<uint16_t*>(a.out)[(size_t)m * a.ldo + row] = f2h<DT>(o);
    return o;
}

// One workgroup = R row tiles (R = 2 for the packed gate/up pair) x all of K; the waves split the 256-wide
// k-blocks; MT = ceil(T/16) M tiles.  GJ (4-bit only) = 32-wide k runs per quantisation group inside a k-block
// (group 32/64/128/>=256 -> 1/2/4/8): the MFMA chain runs over a whole group and the scale is applied once.
// rp (dense3r_kernel: q / k / v of a decode step of 5..32 tokens): RoPE and the cache write in the epilogue; q and k then run as row-tile
// PAIRS (j, j + D/2) of one head, as in dense_small_body
template <int DT, int WTYPE, int MT, int R, int GJ>
__device__ __forceinline__ void dense_body(const DenseArgs& a, const int bx, const DenseRope* rp = nullptr) {
    extern __shared__ __attribute__((aligned(16))) float dg_red[];   // [NW][R][MT][16 m][16 rows]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, NW = blockDim.x >> 6;
    const int r16 = lane & 15, kg = lane >> 4;
    const int nkb = a.K >> 8;
    int row0[R];
    row0[0] = bx * 16;
    if (R == 2) row0[R - 1] = a.pair_offset + bx * 16;
    if (R == 2 && rp) {                                            // RoPE pairs: rows (h, j) and (h, j + D/2) of one head
        const int tph = rp->head_dim >> 5;                         // 16-row pair tiles per head
        row0[0] = (bx / tph) * rp->head_dim + (bx % tph) * 16;
        row0[R - 1] = row0[0] + (rp->head_dim >> 1);
    }
    const uint16_t* x16 = static_cast<const uint16_t*>(a.x);

    f32x4_t y[R][MT];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) y[r][mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // (tried in round 3 for the few-tile 16-bit launches of a batch-32 step: the next k-block's weights and activations requested
    // before the current one is multiplied, two register sets -- 215 VGPRs, wo / down 25.3 -> 28.2 us, q|k|v 22.3 -> 25.5: slower)
    for (int kb = wave; kb < nkb; kb += NW) {
        // ---- all loads of this k-block first: 8 weight fragments per tile, 8 activation fragments per M tile
        uint4 bw[R][8];
        uint4 aw[MT][8];
        if constexpr (WTYPE == DW_DENSE) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                // row-major: row r16 of the tile, 32 j + 8 kg inside the k-block; tiled: fragment j is 64 lanes x 16 B contiguous
                const uint16_t* wp = a.wtiled ? static_cast<const uint16_t*>(a.w) + ((((size_t)(row0[r] >> 4) * nkb + kb) * 8) * 64 + lane) * 8
                                              : static_cast<const uint16_t*>(a.w) + (size_t)(row0[r] + r16) * a.ldw + (size_t)kb * 256 + 8 * kg;
                const int jstride = a.wtiled ? 512 : 32;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const dg_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const dg_u32x4*>(wp + (size_t)j * jstride));
                    bw[r][j] = make_uint4(v.x, v.y, v.z, v.w);
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if constexpr (WTYPE == DW_GPTQ4T) {
                    const dg_u32x4* tp = reinterpret_cast<const dg_u32x4*>(a.w) + ((size_t)(row0[r] >> 4) * nkb + kb) * 128 + lane;
                    const dg_u32x4 v0 = __builtin_nontemporal_load(tp), v1 = __builtin_nontemporal_load(tp + 64);
                    bw[r][0].x = v0.x; bw[r][1].x = v0.y; bw[r][2].x = v0.z; bw[r][3].x = v0.w;
                    bw[r][4].x = v1.x; bw[r][5].x = v1.y; bw[r][6].x = v1.z; bw[r][7].x = v1.w;
                } else {
                    const uint32_t* qp = static_cast<const uint32_t*>(a.w) + (size_t)(kb * 32 + kg) * a.N + row0[r] + r16;
#pragma unroll
                    for (int j = 0; j < 8; ++j) bw[r][j].x = __builtin_nontemporal_load(qp + (size_t)(4 * j) * a.N);
                }
            }
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = min(mt * 16 + r16, a.T - 1);
            const uint16_t* xp = x16 + (size_t)m * a.ldx + (size_t)kb * 256 + 8 * kg;
#pragma unroll
            for (int j = 0; j < 8; ++j) aw[mt][j] = *reinterpret_cast<const uint4*>(xp + 32 * j);
        }
        // ---- contraction
        // (the machine scheduler otherwise sinks every load to just above its MFMA to save registers -- 48-60 VGPRs, eight dependent
        // memory round trips per k-block instead of one; Llama-3-8B bf16 at batch 32: wo 13.2 -> 12.9 us, down 43.9 -> 40.5)
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (WTYPE == DW_DENSE) {
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int j = 0; j < 8; ++j) y[r][mt] = mfma32<DT>(aw[mt][j], bw[r][j], y[r][mt]);
        } else {
            // 128+q (bf16) / 1024+q (f16): OR the 4-bit code into the mantissa of a constant; code pairs
            // (n_i, n_{i+4}) give the element order {0,4,1,5,2,6,3,7} -> permute the activations to match.
            constexpr uint32_t KOFF = (DT == MI355_DTYPE_BF16) ? 0x43004300u : 0x64006400u;
            constexpr float OFF = (DT == MI355_DTYPE_BF16) ? 128.f : 1024.f;
            constexpr uint32_t ONE2 = (DT == MI355_DTYPE_BF16) ? 0x3F803F80u : 0x3C003C00u;
            uint32_t nib = 0x000F000Fu;
            asm volatile("" : "+v"(nib));
            const uint4 ones = make_uint4(ONE2, ONE2, ONE2, ONE2);
            const f32x4_t zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const uint4 v = aw[mt][j];        // v.x=(x0,x1) v.y=(x2,x3) v.z=(x4,x5) v.w=(x6,x7)
                    aw[mt][j].x = __builtin_amdgcn_perm(v.z, v.x, 0x05040100u);   // (x0, x4)
                    aw[mt][j].y = __builtin_amdgcn_perm(v.z, v.x, 0x07060302u);   // (x1, x5)
                    aw[mt][j].z = __builtin_amdgcn_perm(v.w, v.y, 0x05040100u);   // (x2, x6)
                    aw[mt][j].w = __builtin_amdgcn_perm(v.w, v.y, 0x07060302u);   // (x3, x7)
                }
#pragma unroll
            for (int gq = 0; gq < 8 / GJ; ++gq) {
                // groups of 32 / 64 / 128 are exactly GJ runs wide: no run-time division in the loop (GJ == 8: groups of >= 256)
                const int g = GJ < 8 ? kb * (8 / GJ) + gq : (kb * 256) / a.group_size;
                // row sums of x over the group (every column of a ones-MFMA result holds it)
                f32x4_t xs[MT];
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    xs[mt] = zero4;
#pragma unroll
                    for (int jj = 0; jj < GJ; ++jj) xs[mt] = mfma32<DT>(aw[mt][gq * GJ + jj], ones, xs[mt]);
                }
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int col = row0[r] + r16;
                    const float s = h2f<DT>(static_cast<const uint16_t*>(a.scales)[(size_t)g * a.N + marlin_scale_pos(col, a.sperm)]);
                    const float c = -(OFF + zero_point(a, g, col)) * s;
                    f32x4_t acc[MT];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) acc[mt] = zero4;
#pragma unroll
                    for (int jj = 0; jj < GJ; ++jj) {
                        const uint32_t w = bw[r][gq * GJ + jj].x;
                        uint4 b;
                        b.x = (w & nib) | KOFF;
                        b.y = ((w >> 4) & nib) | KOFF;
                        b.z = ((w >> 8) & nib) | KOFF;
                        b.w = ((w >> 12) & nib) | KOFF;
#pragma unroll
                        for (int mt = 0; mt < MT; ++mt) acc[mt] = mfma32<DT>(aw[mt][gq * GJ + jj], b, acc[mt]);
                    }
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int v = 0; v < 4; ++v) y[r][mt][v] = fmaf(s, acc[mt][v], fmaf(c, xs[mt][v], y[r][mt][v]));   // (explicit: the same association in every instantiation)
                }
            }
        }
    }

    // ---- cross-wave reduction: C layout lane (col = weight row r16, rows m = 4kg+v)
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int v = 0; v < 4; ++v)
                dg_red[((((size_t)wave * R + r) * MT + mt) * 16 + 4 * kg + v) * 16 + r16] = y[r][mt][v];
    __syncthreads();
    const int nout = MT * 16 * 16;
    for (int idx = threadIdx.x; idx < nout; idx += blockDim.x) {
        const int rr = idx & 15, m = idx >> 4;
        if (m >= a.T) continue;
        float val[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float s = 0.f;
            for (int w = 0; w < NW; ++w) s += dg_red[((((size_t)w * R + r) * MT + (m >> 4)) * 16 + (m & 15)) * 16 + rr];
            val[r] = s;
        }
        if (rp) {
            // the rounding chain of the projection (+bias), then for q and k "f32 -> rope -> model dtype" (attention.rs:644-690;
            // rope_cache_bf16_kernel's arithmetic), then the cache write of k / v (attention.rs:707-719) -- dense_small_body's epilogue
            const uint16_t* b16 = static_cast<const uint16_t*>(a.bias);
            const int Dh = rp->head_dim, hf = Dh >> 1, bs = rp->block_size, Hkv = rp->n_kv_heads;
            const int64_t slot = rp->slots[m];
            const int64_t blk = slot >= 0 ? slot / bs : 0;
            const int off = slot >= 0 ? (int)(slot % bs) : 0;
            uint16_t* o16 = static_cast<uint16_t*>(a.out);
            float o0 = rnd<DT>(val[0]);
            if (b16) o0 = rnd<DT>(o0 + h2f<DT>(b16[row0[0] + rr]));
            if (R == 2) {
                float o1 = rnd<DT>(val[R - 1]);
                if (b16) o1 = rnd<DT>(o1 + h2f<DT>(b16[row0[R - 1] + rr]));
                const int h = row0[0] / Dh, j = row0[0] % Dh + rr;
                const int64_t pos = rp->positions[m];
                const float cc = rp->cos_t[pos * hf + j], sn = rp->sin_t[pos * hf + j];
                float f0, f1;
                rope_rotate(o0, o1, cc, sn, f0, f1);
                const uint16_t r0 = f2h<DT>(f0), r1 = f2h<DT>(f1);
                o16[(size_t)m * a.ldo + row0[0] + rr] = r0;
                o16[(size_t)m * a.ldo + row0[R - 1] + rr] = r1;
                if (a.rope_mode == 2 && slot >= 0) {
                    const int d0 = j, d1 = j + hf;
                    if (rp->flash) { rp->kcache[(slot * Hkv + h) * Dh + d0] = r0; rp->kcache[(slot * Hkv + h) * Dh + d1] = r1; }
                    else {
                        rp->kcache[((((blk * Hkv + h) * (Dh / 8) + d0 / 8) * bs + off) * 8) + d0 % 8] = r0;
                        rp->kcache[((((blk * Hkv + h) * (Dh / 8) + d1 / 8) * bs + off) * 8) + d1 % 8] = r1;
                    }
                }
            } else {
                const uint16_t r0 = f2h<DT>(o0);
                const int row = row0[0] + rr, h = row / Dh, d = row % Dh;
                o16[(size_t)m * a.ldo + row] = r0;
                if (a.rope_mode == 3 && slot >= 0)
                    rp->vcache[rp->flash ? (slot * Hkv + h) * Dh + d : ((blk * Hkv + h) * Dh + d) * (int64_t)bs + off] = r0;
            }
            continue;
        }
        dense_epilogue<DT>(a, m, row0[0] + rr, val[0], val[R - 1]);
    }
}
template <int DT, int WTYPE, int MT, int R, int GJ>
__global__ void __launch_bounds__(512) dense_kernel(const DenseArgs a) { dense_body<DT, WTYPE, MT, R, GJ>(a, (int)blockIdx.x); }
// q, k and v projections of one attention block in ONE launch (same x, k, tokens and weight format; three outputs): the k and
// v launches of a grouped-query model are 32 workgroups of 6 us each (Qwen2-7B: 2 of the 3 launches for 1/8 of the bytes).
// The workgroup index selects the matrix -- a uniform branch in front of one body.
template <int DT, int WTYPE, int MT, int GJ>
__global__ void __launch_bounds__(512) dense3_kernel(const DenseArgs a0, const DenseArgs a1, const DenseArgs a2, const int t0, const int t1) {
    const int bx = (int)blockIdx.x;
    if (bx < t0) dense_body<DT, WTYPE, MT, 1, GJ>(a0, bx);
    else if (bx < t0 + t1) dense_body<DT, WTYPE, MT, 1, GJ>(a1, bx - t0);
    else dense_body<DT, WTYPE, MT, 1, GJ>(a2, bx - t0 - t1);
}

// the same with RoPE and the cache write in the epilogue (decode steps of 5..32 tokens, bf16 cache, full non-interleaved rotary): q and k
// as pairs of row tiles half a head apart (t0, t1 count PAIRS), v one tile per workgroup -- no rope_cache launch behind it
template <int DT, int WTYPE, int MT, int GJ>
__global__ void __launch_bounds__(512) dense3r_kernel(const DenseArgs a0, const DenseArgs a1, const DenseArgs a2, const int t0, const int t1,
                                                      const DenseRope rp) {
    const int bx = (int)blockIdx.x;
    if (bx < t0) dense_body<DT, WTYPE, MT, 2, GJ>(a0, bx, &rp);
    else if (bx < t0 + t1) dense_body<DT, WTYPE, MT, 2, GJ>(a1, bx - t0, &rp);
    else dense_body<DT, WTYPE, MT, 1, GJ>(a2, bx - t0 - t1, &rp);
}

// ------------------------------------------------------------------------------------------------ 5..64 tokens x 16-bit weights, many row tiles
// dense_body gives a workgroup ONE row tile (or gate/up pair) and lets its waves split K: every wave then fetches its own copy of
// the activations of its k-blocks -- at 32 tokens 16 KB of x (from L2) for every 8 KB of weights, and all of a k-block's loads are
// requested, waited for and consumed before the next k-block's go out (measured on BASELINE configs[2], Llama-3-8B bf16 at batch
// 32: gate/up 235 MB in 86.7 us = 2.7 TB/s, lm_head 2.7 TB/s).  Here the waves of a workgroup own DIFFERENT row tiles and sweep K
// together:
//   * one image of the activations per k-block in LDS, staged once by all threads and shared by the NW tiles (rows padded to
//     528 B: the 16 rows of an A-fragment read land on 16 different bank quads), double-buffered, one barrier per k-block;
//   * every wave keeps TWO k-blocks of its tile's weights in flight (a 2-slot register ring; the activations of the next
//     k-block are requested in between, so the counted vmcnt waits never drain the ring);
//   * a wave owns all of K for its tile: no cross-wave reduction, the epilogue runs out of the accumulators.
// Taken when the launch has enough row tiles for that shape (gate/up, lm_head); the few-tile projections (wo, down, q/k/v) keep
// the K-split of dense_body.
constexpr int DWD_LDX = 264;                        // 16-bit elements per staged row: 256 + 8 pad
//   * round 5: few-tile launches with a long K (down: 256 tiles x 56 k-blocks at Llama-3-8B) split K ACROSS workgroups (gridDim.y, two
//     workgroups per CU) and leave f32 partial sums [split][token][row] for gptq_wide_epilogue_kernel (split order: deterministic) --
//     on dense_body they ran at 2.7 TB/s (one k-block per wave at a time, every wave its own activations).
template <int DT, int MT, int R, int NW>
__global__ void __launch_bounds__(64 * NW) dense_wide_kernel(const DenseArgs a, float* __restrict__ part, const int ldp, const int kb_per_split) {
    extern __shared__ __attribute__((aligned(16))) uint16_t dwd_x[];          // [2][MT * 16][DWD_LDX]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r16 = lane & 15, kg = lane >> 4;
    const int nkb_all = a.K >> 8, T = a.T;
    const int kb_lo = (int)blockIdx.y * kb_per_split, nkb = min(nkb_all, kb_lo + kb_per_split) - kb_lo;   // this workgroup's K range
    if (nkb <= 0) return;
    // every workgroup starts its sweep at a different k-block: in step, all waves of the launch would ask for the same 512-byte column
    // of rows 8 KB apart (the order of the f32 sum changes with the workgroup, not from run to run)
    const int rot = (int)((blockIdx.x * 7u) % (unsigned)nkb);
#define DWD_KB(KB_) (kb_lo + ((KB_) + rot) % nkb)
    const int ntiles = (R == 2 ? a.pair_offset : a.N) >> 4;
    const int tile = min((int)blockIdx.x * NW + wave, ntiles - 1);            // a surplus wave shadows the last tile (stores skipped)
    const bool live = (int)blockIdx.x * NW + wave < ntiles;
    int row0[R];
    row0[0] = tile * 16;
    if (R == 2) row0[R - 1] = a.pair_offset + tile * 16;
    const uint16_t* x16 = static_cast<const uint16_t*>(a.x);
    const uint16_t* wrow[R];
#pragma unroll
    for (int r = 0; r < R; ++r)
        wrow[r] = a.wtiled ? static_cast<const uint16_t*>(a.w) + (((size_t)(row0[r] >> 4) * nkb_all) * 512 + lane) * 8
                           : static_cast<const uint16_t*>(a.w) + (size_t)(row0[r] + r16) * a.ldw + 8 * kg;
    const int wkstride = a.wtiled ? 4096 : 256, wjstride = a.wtiled ? 512 : 32;      // elements per k-block / per fragment

    // ---- staging: MT * 512 pieces of 16 B per k-block over NW * 64 threads (piece p: row p / 32, 16-byte column p % 32)
    constexpr int XP = MT * 8 / NW;
    static_assert(XP >= 1 && XP * NW == MT * 8, "pieces must divide evenly");
    const int xcol = threadIdx.x & 31, xrow0 = threadIdx.x >> 5;             // piece i of this thread: row xrow0 + 2 NW i
    const uint16_t* xsrc[XP];
#pragma unroll
    for (int i = 0; i < XP; ++i) xsrc[i] = x16 + (size_t)min(xrow0 + 2 * NW * i, T - 1) * a.ldx + 8 * xcol;
    uint16_t* const xdst = dwd_x + (size_t)xrow0 * DWD_LDX + 8 * xcol;
    uint4 xr[XP];
#define DWD_X_LOAD(KB_) _Pragma("unroll") for (int i_ = 0; i_ < XP; ++i_) xr[i_] = *reinterpret_cast<const uint4*>(xsrc[i_] + (size_t)DWD_KB(KB_) * 256)
#define DWD_X_STORE(BUF_) _Pragma("unroll") for (int i_ = 0; i_ < XP; ++i_) \
        *reinterpret_cast<uint4*>(xdst + ((size_t)(BUF_) * MT * 16 + 2 * NW * i_) * DWD_LDX) = xr[i_]
    uint4 bw[2][R][8];
    // past the end: one 16-byte request per instruction (see dense_small_body)
#define DWD_W_LOAD(SLOT_, KB_) do { \
        const bool ok_ = (KB_) < nkb; \
        const size_t koff_ = ok_ ? (size_t)DWD_KB(KB_) * wkstride : 0; \
        _Pragma("unroll") for (int r_ = 0; r_ < R; ++r_) { \
            const uint16_t* wp_ = (ok_ ? wrow[r_] : static_cast<const uint16_t*>(a.w)) + koff_; \
            _Pragma("unroll") for (int j_ = 0; j_ < 8; ++j_) { \
                const dg_u32x4 v_ = __builtin_nontemporal_load(reinterpret_cast<const dg_u32x4*>(wp_ + (ok_ ? wjstride * j_ : 0))); \
                bw[SLOT_][r_][j_] = make_uint4(v_.x, v_.y, v_.z, v_.w); \
            } \
        } } while (0)
    DWD_X_LOAD(0);
    DWD_W_LOAD(0, 0);
    DWD_W_LOAD(1, 1);
    f32x4_t y[R][MT];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) y[r][mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    DWD_X_STORE(0);
    __syncthreads();

#define DWD_COMPUTE(SLOT_, BUF_) do { \
        const uint16_t* xb_ = dwd_x + ((size_t)(BUF_) * MT * 16 + r16) * DWD_LDX + 8 * kg; \
        _Pragma("unroll") for (int j_ = 0; j_ < 8; ++j_) { \
            uint4 aw_[MT]; \
            _Pragma("unroll") for (int mt_ = 0; mt_ < MT; ++mt_) aw_[mt_] = *reinterpret_cast<const uint4*>(xb_ + (size_t)mt_ * 16 * DWD_LDX + 32 * j_); \
            _Pragma("unroll") for (int r_ = 0; r_ < R; ++r_) \
                _Pragma("unroll") for (int mt_ = 0; mt_ < MT; ++mt_) y[r_][mt_] = mfma32<DT>(aw_[mt_], bw[SLOT_][r_][j_], y[r_][mt_]); \
        } } while (0)
    for (int kb0 = 0; kb0 < nkb; kb0 += 2) {
        // slot 0
        {
            const int kb = kb0;
            DWD_X_LOAD(min(kb + 1, nkb - 1));                                // next k-block's activations ride between the ring's loads
            DWD_COMPUTE(0, 0);
            DWD_W_LOAD(0, kb + 2);
            DWD_X_STORE(1);
            __syncthreads();
        }
        if (kb0 + 1 < nkb) {                                                 // workgroup-uniform
            const int kb = kb0 + 1;
            DWD_X_LOAD(min(kb + 1, nkb - 1));
            DWD_COMPUTE(1, 1);
            DWD_W_LOAD(1, kb + 2);
            DWD_X_STORE(0);
            __syncthreads();
        }
    }
#undef DWD_X_LOAD
#undef DWD_X_STORE
#undef DWD_W_LOAD
#undef DWD_COMPUTE
#undef DWD_KB
    if (!live) return;
    if (part) {
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int v = 0; v < 4; ++v)
                    part[((size_t)blockIdx.y * (MT * 16) + mt * 16 + 4 * kg + v) * ldp + row0[r] + r16] = y[r][mt][v];
        return;
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int m = mt * 16 + 4 * kg + v;
            if (m < T) dense_epilogue<DT>(a, m, row0[0] + r16, y[0][mt][v], y[R - 1][mt][v]);
        }
}

// ------------------------------------------------------------------------------------------------ 1..4 tokens x 4-bit weights
// The decode step of a GPTQ / AWQ model at batch 1..4 is a chain of small weight streams (Qwen2-7B: 6 / 8 / 34 / 68 MB), each
// against the launch floor (~4.5 us): what a launch reaches is (bytes in flight) / (memory latency), and what the step reaches is
// mostly the NUMBER of launches.  dense_body asks for one k-block per wave at a time and its activation and scale loads queue
// BEHIND the weight loads (VMEM returns in order).  Here:
//   * every wave stages the activations of ITS k-blocks once, into its own LDS region -- already in the (x0,x4,x1,x5,..) order
//     of the 4-bit code pairs -- one round trip together with the first weight loads, no workgroup barrier;
//   * when `norm_w` is set the staging applies RmsNorm: inv = rsqrt(sum(x^2)/K + eps) comes from `ss_in`, per-tile partial sums of
//     squares that the PRODUCER of the residual stream left behind in its epilogue (`ss_out`; fixed summation order on both
//     sides) -- the two norm launches of a layer disappear without a second pass over x (measured first: every workgroup
//     redoing the two-pass norm behind a barrier cost as much as the launches it saved);
//   * every wave keeps D k-blocks of weights + the dwords holding their scales / zero points in flight (a register ring of D
//     slots; the only VMEM instructions in the loop are the ring's loads, so the counted vmcnt waits leave D-1 slots outstanding).
// Same arithmetic as dense_body's 4-bit arm (128+q / 1024+q codes, scale applied once per group, ones-MFMA row sums).
// (wave_sum_dpp, common.h: __shfl_xor reductions measured 6 us on the 28 us gate/up launch when every wave reduced four tokens'
// partial sums that way)
constexpr int DS_X_BYTES_MAX = 40 * 1024;          // staged activations: T * K * 2 bytes (T = 1: K <= 20480)
template <int DT, int WTYPE, int R, int GJ, int D, bool ZP, int NCH>
__device__ __forceinline__ void dense_small_body(const DenseArgs& a, const int bx, const DenseRope* rp = nullptr) {
    static_assert(WTYPE != DW_DENSE, "4-bit weights only");
    constexpr int NG = 8 / GJ;                     // quantisation groups per k-block (GJ == 8: one group of >= 256)
    extern __shared__ __attribute__((aligned(16))) uint8_t ds_lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), NW = blockDim.x >> 6;
    const int r16 = lane & 15, kg = lane >> 4;
    const int nkb = a.K >> 8, T = a.T, K = a.K;
    const int nmine = (nkb - wave + NW - 1) / NW;                  // NW <= nkb: every wave has at least one k-block
    const int nmax = (nkb + NW - 1) / NW;
    uint16_t* xw = reinterpret_cast<uint16_t*>(ds_lds) + (size_t)wave * nmax * T * 256;    // this wave's [nmine][T][256], pair order
    float* red = reinterpret_cast<float*>(ds_lds + (size_t)NW * nmax * T * 512);           // [NW][R][4][16]
    int row0[R];
    row0[0] = bx * 16;
    if (R == 2) row0[R - 1] = a.pair_offset + bx * 16;
    if (R == 2 && rp) {                                            // RoPE pairs: rows (h, j) and (h, j + D/2) of one head
        const int tph = rp->head_dim >> 5;                         // 16-row pair tiles per head
        row0[0] = (bx / tph) * rp->head_dim + (bx % tph) * 16;
        row0[R - 1] = row0[0] + (rp->head_dim >> 1);
    }

    // ---- activation (and norm) loads first, then the ring's first D slots
    const uint16_t* x16 = static_cast<const uint16_t*>(a.x);
    const uint16_t* nw16 = static_cast<const uint16_t*>(a.norm_w);
    // entry f = i * T + t: 256 values = 32 lanes x 16 B, two entries per pass, NCH * 4 passes (the launch checks F <= NCH * 8).
    // Everything below is straight-line code: a run-time loop around loads in front of the ring makes the compiler's wait-count
    // pass give up on counted waits in the main loop (it drains the whole ring at the loop header).
    const int F = nmine * T, half = lane >> 5, piece = lane & 31;
    constexpr int SC = NCH * 4;
    float inv[4] = {1.f, 1.f, 1.f, 1.f};
    // Loads that have nothing to fetch (token >= T, pass beyond F) are still issued -- conditional loads make the wait-count pass
    // drain everything -- but with every lane on the SAME 16 bytes: one 64-byte request instead of a KiB (unpredicated and
    // clamped, a wave of the gate/up launch pulled 16 KB of partial sums, activations and norm weights for its 8 KB of weights).
    f32x4_t ssp[4][2];
    const f32x4_t z4 = {0.f, 0.f, 0.f, 0.f};
    if (nw16) {
        const int ntp = K >> 4;                                          // the producer's tiles (<= 512: K <= 8192 when normed), 4 per lane and load
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const bool ok = t < T && 256 * c < ntp;
                ssp[t][c] = *reinterpret_cast<const f32x4_t*>(a.ss_in + (ok ? (size_t)t * ntp + min(4 * lane + 256 * c, ntp - 4) : 0));   // masked at use
            }
    }
    uint4 xv[SC], gv[SC];
#pragma unroll
    for (int c = 0; c < SC; ++c) {
        const bool ok = 2 * c < F;
        const int f = min(2 * c + half, F - 1), i = f / T, t = f - i * T;
        const size_t k0 = (size_t)(wave + i * NW) * 256 + 8 * piece;
        xv[c] = *reinterpret_cast<const uint4*>(x16 + (ok ? (size_t)t * a.ldx + k0 : 0));
        if (nw16) gv[c] = *reinterpret_cast<const uint4*>(nw16 + (ok ? k0 : 0));
    }

    // sc: the aligned DWORD that holds the 16-bit scale (extracted at use): 16-bit loads get packed two to a VGPR by the compiler
    // right after they are issued -- a wait on the NEWEST loads of the ring, i.e. on all of it
    struct Slot { uint32_t q[R][8]; uint32_t sc[R][NG]; uint32_t zw[R][NG]; };
    int spos[R], zidx[R], zsh[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int col = row0[r] + r16;
        spos[r] = marlin_scale_pos(col, a.sperm);
        int zp = col;
        if (a.zmode == MI355_ZERO_AWQ_MARLIN) {        // marlin zero points (convert_awq_marlin.py:72-91), as zero_point()
            const int p1 = marlin_scale_pos(col, SP_GROUPED), u = p1 & 7;
            zp = (p1 & ~7) + (((u & 1) << 2) | (u >> 1));
        }
        zidx[r] = zp >> 3; zsh[r] = 4 * (zp & 7);
    }
    const float zadd = a.zmode == MI355_ZERO_GPTQ_PLUS1 ? 1.f : 0.f;
    const uint16_t* sc16 = static_cast<const uint16_t*>(a.scales);
    // Past the wave's last k-block the slot is still loaded (conditional loads in the ring make the compiler's wait-count pass drain
    // everything: measured), but every lane reads the SAME 16 bytes of the last block: one 64-byte request instead of a KiB.
    auto issue = [&](Slot& sl, const int kb_in) {
        const bool ok = kb_in < nkb;
        const int kb = ok ? kb_in : nkb - 1, ln = ok ? lane : 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if constexpr (WTYPE == DW_GPTQ4T) {
                const dg_u32x4* tp = reinterpret_cast<const dg_u32x4*>(a.w) + ((size_t)(row0[r] >> 4) * nkb + kb) * 128 + ln;
                const dg_u32x4 v0 = __builtin_nontemporal_load(tp), v1 = __builtin_nontemporal_load(tp + (ok ? 64 : 0));
                sl.q[r][0] = v0.x; sl.q[r][1] = v0.y; sl.q[r][2] = v0.z; sl.q[r][3] = v0.w;
                sl.q[r][4] = v1.x; sl.q[r][5] = v1.y; sl.q[r][6] = v1.z; sl.q[r][7] = v1.w;
            } else {
                const uint32_t* qp = static_cast<const uint32_t*>(a.w) + (size_t)(kb * 32 + (ok ? kg : 0)) * a.N + row0[r] + (ok ? r16 : 0);
#pragma unroll
                for (int j = 0; j < 8; ++j) sl.q[r][j] = __builtin_nontemporal_load(qp + (size_t)(ok ? 4 * j : 0) * a.N);
            }
#pragma unroll
            for (int gq = 0; gq < NG; ++gq) {
                const int g = GJ < 8 ? kb * NG + gq : (kb * 256) / a.group_size;
                sl.sc[r][gq] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(sc16 + (size_t)g * a.N + (ok ? (spos[r] & ~1) : 0)));
                if constexpr (ZP) sl.zw[r][gq] = a.qzeros[(size_t)g * (a.N / 8) + (ok ? zidx[r] : 0)];
            }
        }
    };
    Slot ring[D];
#pragma unroll
    for (int d = 0; d < D; ++d) issue(ring[d], wave + d * NW);

    // ---- activations -> this wave's LDS region (rms_norm_bf16_kernel's arithmetic, elementwise.hip, when normed)
    if (nw16) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (t >= T) break;
            const int ntp = K >> 4;
            const f32x4_t p0 = 4 * lane < ntp ? ssp[t][0] : z4, p1 = 4 * lane + 256 < ntp ? ssp[t][1] : z4;    // (t < T here)
            float v = ((p0[0] + p0[1]) + (p0[2] + p0[3])) + ((p1[0] + p1[1]) + (p1[2] + p1[3]));
            inv[t] = rsqrtf(wave_sum_dpp(v) / (float)K + a.norm_eps);
        }
    }
#pragma unroll
    for (int c = 0; c < SC; ++c) {
        const int f = 2 * c + half;
        if (f < F) {
            uint4 v = xv[c];
            if (nw16) {
                const int t = f % T;
                const float iv = t == 0 ? inv[0] : t == 1 ? inv[1] : t == 2 ? inv[2] : inv[3];
                const uint32_t u[4] = {v.x, v.y, v.z, v.w}, gw[4] = {gv[c].x, gv[c].y, gv[c].z, gv[c].w};
                uint32_t o[4];
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    o[e] = pack_bf16x2(bf16lo_to_f32(u[e]) * iv * bf16lo_to_f32(gw[e]), bf16hi_to_f32(u[e]) * iv * bf16hi_to_f32(gw[e]));
                v = make_uint4(o[0], o[1], o[2], o[3]);
            }
            uint4 pq;                                                        // code pairs (n_i, n_{i+4}): element order {0,4,1,5,2,6,3,7}
            pq.x = __builtin_amdgcn_perm(v.z, v.x, 0x05040100u);
            pq.y = __builtin_amdgcn_perm(v.z, v.x, 0x07060302u);
            pq.z = __builtin_amdgcn_perm(v.w, v.y, 0x05040100u);
            pq.w = __builtin_amdgcn_perm(v.w, v.y, 0x07060302u);
            *reinterpret_cast<uint4*>(xw + (size_t)f * 256 + 8 * piece) = pq;
        }
    }

    constexpr uint32_t KOFF = (DT == MI355_DTYPE_BF16) ? 0x43004300u : 0x64006400u;
    constexpr float OFF = (DT == MI355_DTYPE_BF16) ? 128.f : 1024.f;
    constexpr uint32_t ONE2 = (DT == MI355_DTYPE_BF16) ? 0x3F803F80u : 0x3C003C00u;
    uint32_t nib = 0x000F000Fu;
    asm volatile("" : "+v"(nib));
    const uint4 ones = make_uint4(ONE2, ONE2, ONE2, ONE2);
    const f32x4_t zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4_t y[R];
#pragma unroll
    for (int r = 0; r < R; ++r) y[r] = zero4;
    const uint16_t* xrow = xw + (size_t)min(r16, T - 1) * 256 + 8 * kg;

    auto compute = [&](const Slot& sl, const int i) {
        uint4 aw[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) aw[j] = *reinterpret_cast<const uint4*>(xrow + (size_t)i * T * 256 + 32 * j);
#pragma unroll
        for (int gq = 0; gq < NG; ++gq) {
            f32x4_t xs = zero4;
#pragma unroll
            for (int jj = 0; jj < GJ; ++jj) xs = mfma32<DT>(aw[gq * GJ + jj], ones, xs);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float s = h2f<DT>((uint16_t)(sl.sc[r][gq] >> (16 * (spos[r] & 1))));
                float z = 8.f;
                if constexpr (ZP) z = (float)((sl.zw[r][gq] >> zsh[r]) & 0xFu) + zadd;
                const float c = -(OFF + z) * s;
                f32x4_t acc = zero4;
#pragma unroll
                for (int jj = 0; jj < GJ; ++jj) {
                    const uint32_t w = sl.q[r][gq * GJ + jj];
                    uint4 b;
                    b.x = (w & nib) | KOFF;
                    b.y = ((w >> 4) & nib) | KOFF;
                    b.z = ((w >> 8) & nib) | KOFF;
                    b.w = ((w >> 12) & nib) | KOFF;
                    acc = mfma32<DT>(aw[gq * GJ + jj], b, acc);
                }
#pragma unroll
                for (int v = 0; v < 4; ++v) y[r][v] += s * acc[v] + c * xs[v];
            }
        }
    };
    for (int i0 = 0; i0 < nmine; i0 += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int i = i0 + d;
            if (i < nmine) compute(ring[d], i);
            issue(ring[d], wave + (i + D) * NW);
        }
    }

    // ---- cross-wave reduction: C layout lane (col = weight row r16, rows m = 4kg+v): tokens 0..3 sit in kg == 0
    if (kg == 0) {
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int v = 0; v < 4; ++v) red[((wave * R + r) * 4 + v) * 16 + r16] = y[r][v];
    }
    __syncthreads();
    if ((int)threadIdx.x < T * 16) {
        const int rr = threadIdx.x & 15, m = threadIdx.x >> 4;
        float val[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float sum = 0.f;
            for (int w = 0; w < NW; ++w) sum += red[((w * R + r) * 4 + m) * 16 + rr];
            val[r] = sum;
        }
        if (rp) {
            // q / k / v of a decode step: the rounding chain of the projection (+bias), then for q and k "f32 -> rope -> model
            // dtype" (attention.rs:644-690; rope_cache_bf16_kernel's arithmetic), then the cache write of k / v (attention.rs:707-719)
            const uint16_t* b16 = static_cast<const uint16_t*>(a.bias);
            const int Dh = rp->head_dim, hf = Dh >> 1, bs = rp->block_size, Hkv = rp->n_kv_heads;
            const int64_t slot = rp->slots[m];
            const int64_t blk = slot >= 0 ? slot / bs : 0;
            const int off = slot >= 0 ? (int)(slot % bs) : 0;
            uint16_t* o16 = static_cast<uint16_t*>(a.out);
            float o0 = rnd<DT>(val[0]);
            if (b16) o0 = rnd<DT>(o0 + h2f<DT>(b16[row0[0] + rr]));
            if (R == 2) {
                float o1 = rnd<DT>(val[R - 1]);
                if (b16) o1 = rnd<DT>(o1 + h2f<DT>(b16[row0[R - 1] + rr]));
                const int h = row0[0] / Dh, j = row0[0] % Dh + rr;
                const int64_t pos = rp->positions[m];
                const float cc = rp->cos_t[pos * hf + j], sn = rp->sin_t[pos * hf + j];
                float f0, f1;
                rope_rotate(o0, o1, cc, sn, f0, f1);
                const uint16_t r0 = f2h<DT>(f0), r1 = f2h<DT>(f1);
                o16[(size_t)m * a.ldo + row0[0] + rr] = r0;
                o16[(size_t)m * a.ldo + row0[R - 1] + rr] = r1;
                if (a.rope_mode == 2 && slot >= 0) {
                    const int d0 = j, d1 = j + hf;
                    if (rp->flash) { rp->kcache[(slot * Hkv + h) * Dh + d0] = r0; rp->kcache[(slot * Hkv + h) * Dh + d1] = r1; }
                    else {
                        rp->kcache[((((blk * Hkv + h) * (Dh / 8) + d0 / 8) * bs + off) * 8) + d0 % 8] = r0;
                        rp->kcache[((((blk * Hkv + h) * (Dh / 8) + d1 / 8) * bs + off) * 8) + d1 % 8] = r1;
                    }
                }
            } else {
                const uint16_t r0 = f2h<DT>(o0);
                const int row = row0[0] + rr, h = row / Dh, d = row % Dh;
                o16[(size_t)m * a.ldo + row] = r0;
                if (a.rope_mode == 3 && slot >= 0)
                    rp->vcache[rp->flash ? (slot * Hkv + h) * Dh + d : ((blk * Hkv + h) * Dh + d) * (int64_t)bs + off] = r0;
            }
            return;
        }
        const float o = dense_epilogue<DT>(a, m, row0[0] + rr, val[0], val[R - 1]);
        if (a.ss_out) {                                                  // sum of squares of this tile's 16 outputs of token m (fixed order)
            int q2 = __float_as_int(o * o);                            // row (16-lane) sum by DPP shifts: lane 15 of the row ends up with it
            q2 = __float_as_int(__int_as_float(q2) + __int_as_float(__builtin_amdgcn_update_dpp(0, q2, 0x111, 0xf, 0xf, true)));
            q2 = __float_as_int(__int_as_float(q2) + __int_as_float(__builtin_amdgcn_update_dpp(0, q2, 0x112, 0xf, 0xf, true)));
            q2 = __float_as_int(__int_as_float(q2) + __int_as_float(__builtin_amdgcn_update_dpp(0, q2, 0x114, 0xf, 0xf, true)));
            q2 = __float_as_int(__int_as_float(q2) + __int_as_float(__builtin_amdgcn_update_dpp(0, q2, 0x118, 0xf, 0xf, true)));
            if (rr == 15) a.ss_out[(size_t)m * (a.ldo >> 4) + bx] = __int_as_float(q2);
        }
    }
}
template <int DT, int WTYPE, int R, int GJ, int D, bool ZP, int NCH>
__global__ void __launch_bounds__(512) dense_small_kernel(const DenseArgs a) { dense_small_body<DT, WTYPE, R, GJ, D, ZP, NCH>(a, (int)blockIdx.x); }
template <int DT, int WTYPE, int GJ, int D>
__global__ void __launch_bounds__(512) dense_small3_kernel(const DenseArgs a0, const DenseArgs a1, const DenseArgs a2, const int t0, const int t1) {
    const int bx = (int)blockIdx.x;
    if (bx < t0) dense_small_body<DT, WTYPE, 1, GJ, D, false, 1>(a0, bx);
    else if (bx < t0 + t1) dense_small_body<DT, WTYPE, 1, GJ, D, false, 1>(a1, bx - t0);
    else dense_small_body<DT, WTYPE, 1, GJ, D, false, 1>(a2, bx - t0 - t1);
}

// q, k, v of a decode step with RoPE and the cache write in the epilogue: q and k as row-tile PAIRS (j, j + D/2) of one head
template <int DT, int WTYPE, int GJ>
__global__ void __launch_bounds__(512) dense_small3r_kernel(const DenseArgs a0, const DenseArgs a1, const DenseArgs a2, const int t0, const int t1,
                                                            const DenseRope rp) {
    const int bx = (int)blockIdx.x;
    if (bx < t0) dense_small_body<DT, WTYPE, 2, GJ, 2, false, 1>(a0, bx, &rp);
    else if (bx < t0 + t1) dense_small_body<DT, WTYPE, 2, GJ, 2, false, 1>(a1, bx - t0, &rp);
    else dense_small_body<DT, WTYPE, 1, GJ, 4, false, 1>(a2, bx - t0 - t1, &rp);
}

// ------------------------------------------------------------------------------------------------ launch
template <int DT, int WTYPE, int MT, int R>
static int dense_launch_gj(const DenseArgs& a, int gj, int nw, hipStream_t st) {
    const int tiles = (a.epi == MI355_EPI_SILU_MUL ? a.pair_offset : a.N) / 16;
    const size_t lds = (size_t)nw * R * MT * 256 * sizeof(float);
    dim3 grid(tiles), block(nw * 64);
    if constexpr (WTYPE == DW_DENSE) {
        hipLaunchKernelGGL((dense_kernel<DT, WTYPE, MT, R, 8>), grid, block, lds, st, a);
    } else {
        switch (gj) {
            case 1: hipLaunchKernelGGL((dense_kernel<DT, WTYPE, MT, R, 1>), grid, block, lds, st, a); break;
            case 2: hipLaunchKernelGGL((dense_kernel<DT, WTYPE, MT, R, 2>), grid, block, lds, st, a); break;
            case 4: hipLaunchKernelGGL((dense_kernel<DT, WTYPE, MT, R, 4>), grid, block, lds, st, a); break;
            default: hipLaunchKernelGGL((dense_kernel<DT, WTYPE, MT, R, 8>), grid, block, lds, st, a); break;
        }
    }
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ---- the 1..4-token launch of 4-bit weights (dense_small_kernel): -4 = this shape stays on dense_kernel
static int g_tune_gemm_min_t = 96;     // tuning key 36: tokens from which a 16-bit projection takes the 128 x 128 MFMA GEMM
static int g_tune_gptq_gemm_off = 0;   // tuning key 39: 1 = 4-bit prompt steps run the decode kernel in 64-token chunks (A/B)
static int g_tune_wide_nw = 4;         // tuning key 38: waves (row tiles) per workgroup of dense_wide_kernel: 2 | 4
static int g_tune_wide_off = 0;        // tuning key 37: 1 = 16-bit launches with many row tiles stay on dense_kernel (A/B)
static int g_tune_small_nw = 0;        // tuning key 30: waves per workgroup (0 = chosen from the k-blocks)
static int g_tune_small_nw_big = 0;    // tuning key 35: the same for K > 8192
static int g_tune_gemm_tall = 0;       // tuning key 30 bit 32 (round 4; measured SLOWER, stays off: bf16 prompt step of Llama-3-8B 53.7 k -> 51.5 k tok/s): the 256-token tile of the 16-bit prompt GEMM where it fills the chip twice
static int g_tune_small_off = 0;       // tuning key 31: 1 = keep 1..4 tokens on dense_kernel (A/B measurements)
static int g_tune_gptq_wide_nosplit = 0;   // tuning key 30 bit 256: gate/up pairs of the 5..32-token 4-bit launches stay on one wave each (A/B)
static int g_tune_gptq_wide_off = 0;   // tuning key 30 bit 128: 1 = 5..64-token launches of tiled 4-bit weights stay on dense_kernel (A/B; round 5)
static int g_tune_small_norope = 0;    // tuning key 34: 1 = RoPE and the cache write stay in their own launch
static int g_tune_small_nonorm = 0;    // tuning key 32: 1 = never norm on the way in (the host layer launches the norm separately)
// tuning key 30 (product): bit mask of folded launches switched OFF -- 1 = the 1..4-token 4-bit kernel, 2 = no norm on the way in,
// 4 = RoPE and the cache write in their own launch, 8 = the LDS-shared-activation 16-bit kernel, 16 = the one-pass 4-bit prompt GEMM;
// 32 (switches ON) = the 256-token tile of the 16-bit prompt GEMM where it fills the chip twice, 32 | 64 = wherever that GEMM runs (tests).
// Probe builds: 33 / 35 waves per workgroup of the 1..4-token kernel (hidden-sized K / long K), 36 tokens from which 16-bit
// projections take the MFMA GEMM, 38 waves per workgroup of the LDS-shared-activation kernel.
void mi355_dense_set_small(int key, int v) {
    if (key == 30) {
        g_tune_small_off = v & 1; g_tune_small_nonorm = (v >> 1) & 1; g_tune_small_norope = (v >> 2) & 1;
        g_tune_wide_off = (v >> 3) & 1; g_tune_gptq_gemm_off = (v >> 4) & 1; g_tune_gemm_tall = (v >> 5) & 3;
        g_tune_gptq_wide_off = (v >> 7) & 1;
        g_tune_gptq_wide_nosplit = (v >> 8) & 1;
    }
    else if (key == 33) g_tune_small_nw = v;
    else if (key == 35) g_tune_small_nw_big = v;
    else if (key == 36 && v > 0) g_tune_gemm_min_t = v;
    else if (key == 38) g_tune_wide_nw = v;
}
static inline int dense_small_gj(int group_size) {
    if (group_size >= 256) return (group_size % 256) ? -1 : 8;
    return group_size == 128 ? 4 : group_size == 64 ? 2 : group_size == 32 ? 1 : -1;
}
static inline int dense_small_nw(int nkb) {
    if (g_tune_small_nw >= 1 && g_tune_small_nw <= 8 && g_tune_small_nw <= nkb && nkb <= 32) return g_tune_small_nw;   // experiments: hidden-sized K only
    if (g_tune_small_nw_big >= 1 && g_tune_small_nw_big <= 8 && nkb > 32) return g_tune_small_nw_big;
    if (nkb < 4) return nkb;
    // hidden-sized K (<= 16 k-blocks: 4096): four waves -- more, smaller workgroups per CU balance the launch better than more
    // waves on fewer (Qwen2-7B gate/up, 1184 workgroups: 19.5 us with 2..4 waves, 22.3 us with 7; q/k/v and wo within 0.4 us)
    if (nkb <= 16) return 4;
    int best = 1, best_slots = 1 << 30;
    for (int nw = 8; nw >= 4; --nw) {                               // fewest idle wave-slots, the larger workgroup on a tie
        const int slots = (nkb + nw - 1) / nw * nw;
        if (slots < best_slots) { best_slots = slots; best = nw; }
    }
    return best;
}
// staging passes a wave needs: ceil(k-blocks per wave * T / 8); the kernel is built for 1 and 2
static inline int dense_small_nch(const DenseArgs& a) {
    const int nkb = a.K >> 8, nw = dense_small_nw(nkb);
    return (((nkb + nw - 1) / nw) * a.T + 7) / 8;
}
static inline bool dense_small_ok(const DenseArgs& a, int dt) {
    if (g_tune_small_off || a.T < 1 || a.T > 4 || (a.N & 15) || (a.K & 255) || (size_t)a.T * a.K * 2 > (size_t)DS_X_BYTES_MAX) return false;
    if (a.norm_w && (dt != MI355_DTYPE_BF16 || g_tune_small_nonorm || !a.ss_in || a.K > 8192)) return false;
    if (((uintptr_t)a.x & 15) || (a.ldx & 7)) return false;
    return dense_small_gj(a.group_size) > 0 && dense_small_nch(a) <= 2;
}
template <int DT, int R, int D, bool ZP, int NCH>
static int dense_small_launch_gj(const DenseArgs& a, int gj, hipStream_t st) {
    const int tiles = (R == 2 ? a.pair_offset : a.N) / 16;
    const int nw = dense_small_nw(a.K >> 8);
    const size_t lds = (size_t)nw * (((a.K >> 8) + nw - 1) / nw) * a.T * 512 + (size_t)nw * R * 64 * sizeof(float);
    dim3 grid(tiles), block(nw * 64);
    switch (gj) {
        case 1: hipLaunchKernelGGL((dense_small_kernel<DT, DW_GPTQ4T, R, 1, D, ZP, NCH>), grid, block, lds, st, a); break;
        case 2: hipLaunchKernelGGL((dense_small_kernel<DT, DW_GPTQ4T, R, 2, D, ZP, NCH>), grid, block, lds, st, a); break;
        case 4: hipLaunchKernelGGL((dense_small_kernel<DT, DW_GPTQ4T, R, 4, D, ZP, NCH>), grid, block, lds, st, a); break;
        default: hipLaunchKernelGGL((dense_small_kernel<DT, DW_GPTQ4T, R, 8, D, ZP, NCH>), grid, block, lds, st, a); break;
    }
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
template <int DT, int R, int D, bool ZP>
static int dense_small_launch_nch(const DenseArgs& a, int gj, hipStream_t st) {
    return dense_small_nch(a) <= 1 ? dense_small_launch_gj<DT, R, D, ZP, 1>(a, gj, st) : dense_small_launch_gj<DT, R, D, ZP, 2>(a, gj, st);
}
// the tiled weight image only: the checkpoint layout (mi355_gptq_linear) stays on dense_kernel
template <int DT, int WTYPE>
static int dense_small_launch(const DenseArgs& a, hipStream_t st) {
    if constexpr (WTYPE != DW_GPTQ4T) return -4;
    else {
        if (!dense_small_ok(a, DT)) return -4;
        const int gj = dense_small_gj(a.group_size);
        const bool zp = a.zmode != MI355_ZERO_SYM8 && a.qzeros;
        if (a.epi == MI355_EPI_SILU_MUL)
            return zp ? dense_small_launch_nch<DT, 2, 2, true>(a, gj, st) : dense_small_launch_nch<DT, 2, 2, false>(a, gj, st);
        return zp ? dense_small_launch_nch<DT, 1, 4, true>(a, gj, st) : dense_small_launch_nch<DT, 1, 4, false>(a, gj, st);
    }
}
template <int DT, int WTYPE>
static int dense_small3_launch(const DenseArgs (&a)[3], const DenseRope* rp, hipStream_t st) {
    if constexpr (WTYPE != DW_GPTQ4T) return -4;
    else {
        for (int i = 0; i < 3; ++i)
            if (!dense_small_ok(a[i], DT) || a[i].zmode != MI355_ZERO_SYM8 || dense_small_nch(a[i]) != 1) return -4;
        const int gj = dense_small_gj(a[0].group_size);
        const int nw = dense_small_nw(a[0].K >> 8);
        if (rp) {
            // RoPE + cache write in the epilogue: whole heads of an even, 32-divisible width (pairs of 16-row tiles half a head apart)
            if (g_tune_small_norope || DT != MI355_DTYPE_BF16 || (rp->head_dim & 31) || a[0].N % rp->head_dim || a[1].N != rp->n_kv_heads * rp->head_dim ||
                a[2].N != a[1].N)
                return -4;
            DenseArgs b[3] = {a[0], a[1], a[2]};
            b[0].rope_mode = 1; b[1].rope_mode = 2; b[2].rope_mode = 3;
            const int t0 = a[0].N / 32, t1 = a[1].N / 32, t2 = a[2].N / 16;
            const size_t lds = (size_t)nw * (((a[0].K >> 8) + nw - 1) / nw) * a[0].T * 512 + (size_t)nw * 2 * 64 * sizeof(float);
            dim3 grid(t0 + t1 + t2), block(nw * 64);
#define DS3R(GJ_) hipLaunchKernelGGL((dense_small3r_kernel<DT, DW_GPTQ4T, GJ_>), grid, block, lds, st, b[0], b[1], b[2], t0, t1, *rp)
            switch (gj) { case 1: DS3R(1); break; case 2: DS3R(2); break; case 4: DS3R(4); break; default: DS3R(8); break; }
#undef DS3R
            return hipGetLastError() == hipSuccess ? 0 : -1;
        }
        const int t0 = a[0].N / 16, t1 = a[1].N / 16, t2 = a[2].N / 16;
        const size_t lds = (size_t)nw * (((a[0].K >> 8) + nw - 1) / nw) * a[0].T * 512 + (size_t)nw * 64 * sizeof(float);
        dim3 grid(t0 + t1 + t2), block(nw * 64);
#define DS3(GJ_) hipLaunchKernelGGL((dense_small3_kernel<DT, DW_GPTQ4T, GJ_, 4>), grid, block, lds, st, a[0], a[1], a[2], t0, t1)
        switch (gj) { case 1: DS3(1); break; case 2: DS3(2); break; case 4: DS3(4); break; default: DS3(8); break; }
#undef DS3
        return hipGetLastError() == hipSuccess ? 0 : -1;
    }
}

#include "gptq_wide.inc"   // 5..64 tokens x tiled 4-bit weights: LDS-shared activations, split-K for few-tile launches
template <int DT, int WTYPE>
static int dense_launch_dt(const DenseArgs& a, hipStream_t st) {
    if (a.T < 1 || a.T > 64 || (a.N & 15) || (a.K & 255)) return -2;
    {
        const int rs = dense_small_launch<DT, WTYPE>(a, st);
        if (rs != -4) return rs;
        if (a.norm_w) return -4;                   // only the small kernel norms on the way in: the caller norms separately
    }
    const bool pair = a.epi == MI355_EPI_SILU_MUL;
    const int mt = (a.T + 15) / 16;
    if constexpr (WTYPE == DW_GPTQ4T) {
        const int rw = gptq_wide_launch<DT>(a, st);
        if (rw != -4) return rw;
    }
    if constexpr (WTYPE == DW_DENSE) {
        // enough row tiles for one tile (or gate/up pair) per wave on every CU: the LDS-shared-activation sweep.  Round 5: ALSO the few-tile
        // launches with a long K (down: 256 tiles x 56 k-blocks at Llama-3-8B), split over K until two workgroups sit on every CU; their
        // partial sums meet in gptq_wide_epilogue_kernel (short-K few-tile launches -- wo, q / k / v -- keep dense_kernel: a sweep of 4
        // k-blocks plus the epilogue launch is not faster than its one launch)
        const int wtiles = (pair ? a.pair_offset : a.N) / 16;
        const int nkb_w = a.K >> 8;
        const bool many = wtiles >= 4 * 192, longk = !pair && nkb_w >= 32 && mt <= 3;
        if (!g_tune_wide_off && a.T > 4 && (many || longk) && (!pair || mt <= 2) && !((uintptr_t)a.x & 15) && !(a.ldx & 7) &&
            !((uintptr_t)a.w & 15) && (a.wtiled || !(a.ldw & 7))) {
            const int nwv = many ? (g_tune_wide_nw == 2 ? 2 : 4) : 4;
            const int gx = (wtiles + nwv - 1) / nwv;
            int ks = 1;
            if (!many) {
                g_num_cus_dg = mi355_num_cus();                         // per device (common.h)
                ks = (2 * g_num_cus_dg + gx - 1) / gx;
                if (ks > nkb_w / 4) ks = nkb_w / 4;
                if (ks < 1) ks = 1;
            }
            const int kps = (nkb_w + ks - 1) / ks;
            ks = (nkb_w + kps - 1) / kps;
            float* part = nullptr;
            const int tpad = mt * 16, ldp = a.N;
            if (ks > 1) {
                void* pp = nullptr;
                if (mi355_scratch_get(&pp, MI355_SCR_GPTQ_WIDE, (size_t)ks * tpad * ldp * sizeof(float), st, false)) return -1;
                part = static_cast<float*>(pp);
            }
            const dim3 grid(gx, ks), block(64 * nwv);
            const size_t lds = (size_t)2 * mt * 16 * DWD_LDX * 2;
#define DWD_GO2(MT_, R_, NW_) do { \
                static bool attr = false; \
                if (!attr) { (void)hipFuncSetAttribute((const void*)dense_wide_kernel<DT, MT_, R_, NW_>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024); attr = true; } \
                hipLaunchKernelGGL((dense_wide_kernel<DT, MT_, R_, NW_>), grid, block, lds, st, a, part, ldp, kps); } while (0)
#define DWD_GO(MT_, R_) do { if (nwv == 2) DWD_GO2(MT_, R_, 2); else DWD_GO2(MT_, R_, 4); } while (0)
            if (pair) { if (mt == 1) DWD_GO(1, 2); else DWD_GO(2, 2); }
            else switch (mt) { case 1: DWD_GO(1, 1); break; case 2: DWD_GO(2, 1); break; case 3: DWD_GO(3, 1); break; default: DWD_GO(4, 1); break; }
#undef DWD_GO2
#undef DWD_GO
            if (hipGetLastError() != hipSuccess) return -1;
            if (ks > 1) {
                hipLaunchKernelGGL((gptq_wide_epilogue_kernel<DT>), dim3((a.N + 255) / 256, a.T), dim3(256), 0, st, a, part, ldp, ks, tpad);
                if (hipGetLastError() != hipSuccess) return -1;
            }
            return 0;
        }
    }
    int gj = 8;
    if (WTYPE != DW_DENSE) {
        if (a.group_size >= 256) { if (a.group_size % 256) return -2; gj = 8; }
        else if (a.group_size == 128) gj = 4;
        else if (a.group_size == 64) gj = 2;
        else if (a.group_size == 32) gj = 1;
        else return -2;
    }
    // waves per workgroup: split K so that tiles * waves covers the chip several times, capped by the k-blocks
    const int tiles = (pair ? a.pair_offset : a.N) / 16;
    const int nkb = a.K >> 8;
    int nw = 8;
    while (nw > 1 && (nw > nkb || (size_t)tiles * nw > 256 * 32)) nw >>= 1;
    if (mt >= 3 && nw > 4) nw = 4;
#define DG_CASE(MT_, R_) return dense_launch_gj<DT, WTYPE, MT_, R_>(a, gj, nw, st)
    if (pair) {
        switch (mt) { case 1: DG_CASE(1, 2); case 2: DG_CASE(2, 2); default: return -3; }
    }
    switch (mt) { case 1: DG_CASE(1, 1); case 2: DG_CASE(2, 1); case 3: DG_CASE(3, 1); default: DG_CASE(4, 1); }
#undef DG_CASE
}

template <int DT, int WTYPE>
static int dense3_launch_dt(const DenseArgs (&a)[3], const DenseRope* rp, hipStream_t st) {
    {
        const int rs = dense_small3_launch<DT, WTYPE>(a, rp, st);
        if (rs != -4) return rs;
        if (a[0].norm_w) return -4;                // only the 1..4-token kernel norms on the way in
    }
    const int mt = (a[0].T + 15) / 16;
    if (rp) {
        // RoPE + cache write in the epilogue of the 5..32-token kernel: whole heads of an even, 32-divisible width, bf16
        if (g_tune_small_norope || DT != MI355_DTYPE_BF16 || mt > 2 || (rp->head_dim & 31) || a[0].N % rp->head_dim ||
            a[1].N != rp->n_kv_heads * rp->head_dim || a[2].N != a[1].N)
            return -4;
    }
    int gj = 8;
    if (WTYPE != DW_DENSE) {
        const int g = a[0].group_size;
        if (g >= 256) { if (g % 256) return -2; }
        else if (g == 128) gj = 4;
        else if (g == 64) gj = 2;
        else if (g == 32) gj = 1;
        else return -2;
    }
    const int t0 = a[0].N / 16, t1 = a[1].N / 16, t2 = a[2].N / 16, tiles = t0 + t1 + t2;
    const int nkb = a[0].K >> 8;
    int nw = 8;
    while (nw > 1 && (nw > nkb || (size_t)tiles * nw > 256 * 32)) nw >>= 1;
    if (mt >= 3 && nw > 4) nw = 4;
    if (rp) {
        DenseArgs b[3] = {a[0], a[1], a[2]};
        b[0].rope_mode = 1; b[1].rope_mode = 2; b[2].rope_mode = 3;
        const int p0 = t0 / 2, p1 = t1 / 2;
        const size_t ldr = (size_t)nw * 2 * mt * 256 * sizeof(float);
        dim3 gridr(p0 + p1 + t2), block(nw * 64);
#define DG3R(MT_, GJ_) hipLaunchKernelGGL((dense3r_kernel<DT, WTYPE, MT_, GJ_>), gridr, block, ldr, st, b[0], b[1], b[2], p0, p1, *rp)
#define DG3R_MT(GJ_) if (mt == 1) DG3R(1, GJ_); else DG3R(2, GJ_);
        if (WTYPE == DW_DENSE) { DG3R_MT(8) }
        else switch (gj) { case 1: DG3R_MT(1) break; case 2: DG3R_MT(2) break; case 4: DG3R_MT(4) break; default: DG3R_MT(8) break; }
#undef DG3R_MT
#undef DG3R
        return hipGetLastError() == hipSuccess ? 0 : -1;
    }
    const size_t lds = (size_t)nw * mt * 256 * sizeof(float);
    dim3 grid(tiles), block(nw * 64);
#define DG3(MT_, GJ_) hipLaunchKernelGGL((dense3_kernel<DT, WTYPE, MT_, GJ_>), grid, block, lds, st, a[0], a[1], a[2], t0, t1)
#define DG3_MT(GJ_) switch (mt) { case 1: DG3(1, GJ_); break; case 2: DG3(2, GJ_); break; case 3: DG3(3, GJ_); break; default: DG3(4, GJ_); break; }
    if (WTYPE == DW_DENSE) { DG3_MT(8) }
    else switch (gj) { case 1: DG3_MT(1) break; case 2: DG3_MT(2) break; case 4: DG3_MT(4) break; default: DG3_MT(8) break; }
#undef DG3_MT
#undef DG3
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

template <int DT>
static int dense_dispatch_dt(const DenseArgs& a, int wtype, hipStream_t st) {
    if (wtype == DW_DENSE) return dense_launch_dt<DT, DW_DENSE>(a, st);
    if (wtype == DW_GPTQ4T) return dense_launch_dt<DT, DW_GPTQ4T>(a, st);
    return dense_launch_dt<DT, DW_GPTQ4>(a, st);
}
static int dense_dispatch(const DenseArgs& a, int wtype, int dt, hipStream_t st) {
    if (dt == MI355_DTYPE_BF16) return dense_dispatch_dt<MI355_DTYPE_BF16>(a, wtype, st);
    if (dt == MI355_DTYPE_F16) return dense_dispatch_dt<MI355_DTYPE_F16>(a, wtype, st);
    return -2;
}

// ---- prompt steps (T >= 96, dense 16-bit weights): the matmul is compute-bound -> a hand-written MFMA GEMM
// (`Linear::forward` on prompt chunks, linear.rs:124-172; packed gate_up of mlp.rs:324-352,440-458).  Both operands are
// k-contiguous ("TN": x [T,K], W [N,K]), so a 16 x 16 x 32 A / B fragment is 16 contiguous bytes of a row:
//   * workgroup tile 128 tokens x 128 weight rows, 4 waves as 2 x 2, wave tile 64 x 64 (4 x 4 accumulators), K in steps of 64;
//   * both tiles go global -> LDS by DMA (`global_load_lds_dwordx4`, inline asm so that hipcc neither counts nor drains it:
//     the DMA of K-step t+1 is in flight while step t is multiplied; one counted wait + one barrier per step), two LDS buffers;
//   * LDS rows are 128 B (64 k): the 16-byte slot c of row r is stored at slot c ^ (r & 7) -- the swizzle is applied to the
//     lane's GLOBAL source address (the DMA destination is lane-linear), fragment reads are conflict-free ds_read_b128;
//   * candle's rounding chain in the epilogue, from the f32 accumulators: round to the dtype, + bias (rounded), then
//     + residual (rounded) or silu(gate) rounded * up rounded -- no separate chain kernel, no f32 / 16-bit staging of y;
//   * SILU_MUL: the B tile holds 64 gate rows and the 64 matching up rows (pair_offset apart in W), every wave owns both halves
//     of its 32 output columns, so gate and up meet in registers.
// rocBLAS is no longer used on this path (round 2 called rocblas_gemm_ex here).
template <int NP>
__device__ __forceinline__ void dgemm_dma(const uint8_t* gsrc_lane, uint32_t lds_dst) {
    // NP x 1 KiB: lane l copies 16 B from its own source address (+ i * row_stride handled by the caller through separate calls)
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc_lane), "s"(lds_dst) : "memory");
}

#define DG_BM 128
#define DG_BN 128
#define DG_BK 64
// WR = wave rows: 2 = 128 tokens x 128 weight rows per workgroup (4 waves, two workgroups per CU), 4 = 256 x 128 (8 waves, one workgroup
// per CU; round 4): at 128 x 128 the tile moves 32 KiB through L2 -> LDS per 2.1 MFLOP -- 15 B per kFLOP, ~10 TB/s at the measured 650
// TFLOP/s, which is what the L2 -> CU path carried in every other measurement of this chip -- the taller tile moves 11.4.  Measured (round 4,
// one box, alternated, T = 2048): 53.7 k tok/s with the 128 x 128 tile, 51.5 k with 256 x 128: L2 traffic is not what bounds it; the tall tile
// stays a tested option (tuning key 30 bit 32; bit-identical outputs, tests/test_gpu_linear.py).  A ring of three K steps behind counted
// waits (one workgroup per CU then) was measured too and removed: 53.0 k -> 41.2 k at 128 x 128, 47.5 k at 256 x 128 -- this GEMM lives on its
// two workgroups per CU, not on prefetch depth.
template <int DT, int WR = 2>
__global__ void __launch_bounds__(128 * WR, WR == 2 ? 2 : 1) dense_gemm_kernel(const DenseArgs a) {
    constexpr int BM = 64 * WR, NTHR = 128 * WR, AROUNDS = BM / (NTHR / 8), BROUNDS = DG_BN / (NTHR / 8);   // rows per staging round = NTHR / 8
    constexpr uint32_t A_BYTES = BM * 128u, STAGE = A_BYTES + 16384u;
    extern __shared__ __attribute__((aligned(1024))) uint8_t smem[];   // [2 buffers][A 16 / 32 KiB | B 16 KiB]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int m16 = lane & 15, kg = lane >> 4;
    const bool silu = a.epi == MI355_EPI_SILU_MUL;
    // token tiles fastest: the workgroups that run together share their weight tile through L2
    const int bm = blockIdx.x, bn = blockIdx.y;
    const int t0 = bm * BM;
    const int n_cols = silu ? a.pair_offset : a.N;                      // output columns
    const int c0 = bn * (silu ? 64 : DG_BN);                            // first output column of this tile
    // weight row held by LDS B-row j
    auto wrow = [&](int j) {
        int n = silu ? (j < 64 ? c0 + j : a.pair_offset + c0 + (j - 64)) : c0 + j;
        const int lim = silu ? (j < 64 ? a.pair_offset - 1 : a.N - 1) : a.N - 1;
        return n > lim ? lim : n;                                       // clamped: columns beyond the matrix are computed and never stored
    };
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(__attribute__((address_space(3))) void*)smem);
    const uint8_t* xb = static_cast<const uint8_t*>(a.x);
    const uint8_t* wb = static_cast<const uint8_t*>(a.w);
    // staging: a round covers NTHR / 8 LDS rows (8 threads = the 8 slots of a row; one wave = 8 rows = 1 KiB); this thread's row / slot inside
    // the round.  A takes AROUNDS rounds (BM rows), B BROUNDS (128 rows)
    constexpr int RROWS = NTHR / 8, NWV = NTHR / 64;
    const int srow = tid >> 3, sslot = tid & 7;
    const uint8_t* asrc[AROUNDS];
    const uint8_t* bsrc[BROUNDS];
#pragma unroll
    for (int r = 0; r < AROUNDS; ++r) {
        const int row = RROWS * r + srow;
        int t = t0 + row;
        if (t > a.T - 1) t = a.T - 1;
        asrc[r] = xb + ((size_t)t * a.ldx) * 2 + (size_t)((sslot ^ (row & 7)) * 16);
    }
#pragma unroll
    for (int r = 0; r < BROUNDS; ++r) {
        const int row = RROWS * r + srow;
        if (a.wtiled) {
            // the tiled image IS the fragment order: the B tile in LDS is a verbatim copy -- [n-tile 8][fragment 2][lane 64][16 B] --
            // and every DMA instruction of a wave copies one contiguous KiB (piece p = NWV r + wave: n-tile p / 2, fragment p % 2 of
            // this K step; 2 KiB per (n-tile, K step)).  (A first version kept the row-major LDS layout and gathered 16-byte chunks:
            // 64 separate requests per instruction, prompt step 40.2 k -> 36.9 k tok/s.)
            const int p = NWV * r + wave;
            const int n = wrow(16 * (p >> 1));
            bsrc[r] = wb + (((((size_t)(n >> 4) * (a.K >> 8)) * 8 + (p & 1)) * 64 + lane) << 4);
        } else {
            bsrc[r] = wb + ((size_t)wrow(row) * a.ldw) * 2 + (size_t)((sslot ^ (row & 7)) * 16);
        }
    }
    const size_t bstep = a.wtiled ? 2048 : DG_BK * 2;                   // bytes per K step on the weight side
    auto stage = [&](int kt, int buf) {
        const uint32_t base = lds0 + (uint32_t)buf * STAGE + (uint32_t)wave * 1024u;
#pragma unroll
        for (int r = 0; r < AROUNDS; ++r) dgemm_dma<1>(asrc[r] + (size_t)kt * (DG_BK * 2), base + (uint32_t)r * (uint32_t)(RROWS * 128));
#pragma unroll
        for (int r = 0; r < BROUNDS; ++r) dgemm_dma<1>(bsrc[r] + (size_t)kt * bstep, base + A_BYTES + (uint32_t)r * (uint32_t)(RROWS * 128));
    };
    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // LDS byte offsets of this lane's fragments (without the K-step slot): A rows of the wave's four m-fragments, B rows of its
    // four n-fragments (SILU_MUL: fragments 0, 1 = gate columns, 2, 3 = the same columns of up)
    int arow[4], brow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) arow[i] = wr * 64 + i * 16 + m16;
#pragma unroll
    for (int j = 0; j < 4; ++j) brow[j] = silu ? ((j >> 1) * 64 + wc * 32 + (j & 1) * 16 + m16) : (wc * 64 + j * 16 + m16);
    const int nkt = a.K / DG_BK;
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nkt) stage(kt + 1, buf ^ 1);                       // in flight while this step is multiplied
        const uint8_t* A = smem + (size_t)buf * STAGE;
        const uint8_t* B = A + A_BYTES;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            uint4 af[4], bf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const uint4*>(A + (size_t)arow[i] * 128 + (size_t)(((s2 * 4 + kg) ^ (arow[i] & 7)) * 16));
#pragma unroll
            for (int j = 0; j < 4; ++j)
                bf[j] = a.wtiled ? *reinterpret_cast<const uint4*>(B + (size_t)((((brow[j] >> 4) * 2 + s2) * 64 + kg * 16 + m16) * 16))
                                 : *reinterpret_cast<const uint4*>(B + (size_t)brow[j] * 128 + (size_t)(((s2 * 4 + kg) ^ (brow[j] & 7)) * 16));
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = mfma32<DT>(af[i], bf[j], acc[i][j]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // the next tile has landed (this wave's share of it)
        __syncthreads();                                                // ... everyone's; and everyone is done reading `buf`
    }
    // ---- epilogue: the lane holds tokens 4 kg + v of every m-fragment for column m16 of every n-fragment
    const uint16_t* bias = static_cast<const uint16_t*>(a.bias);
    const uint16_t* resid = static_cast<const uint16_t*>(a.resid);
    uint16_t* out = static_cast<uint16_t*>(a.out);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int t = t0 + wr * 64 + i * 16 + 4 * kg + v;
            if (t >= a.T) continue;
            if (silu) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int col = c0 + wc * 32 + j * 16 + m16;
                    if (col >= n_cols) continue;
                    float g = rnd<DT>(acc[i][j][v]), u = rnd<DT>(acc[i][j + 2][v]);
                    if (bias) { g = rnd<DT>(g + h2f<DT>(bias[col])); u = rnd<DT>(u + h2f<DT>(bias[a.pair_offset + col])); }
                    out[(size_t)t * a.ldo + col] = f2h<DT>(rnd<DT>(rnd<DT>(g / (1.f + __expf(-g))) * u));      // silu(gate) * up (mlp.rs:457)
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int col = c0 + wc * 64 + j * 16 + m16;
                    if (col >= n_cols) continue;
                    float o = rnd<DT>(acc[i][j][v]);
                    if (bias) o = rnd<DT>(o + h2f<DT>(bias[col]));
                    if (a.epi == MI355_EPI_RESID) o = rnd<DT>(o + h2f<DT>(resid[(size_t)t * a.ldo + col]));
                    out[(size_t)t * a.ldo + col] = f2h<DT>(o);
                }
            }
        }
}

// ---- prompt steps over the tiled 4-bit image: ONE pass over the weights instead of T / 64 passes of the decode
// kernel.  Same 128 x 128 workgroup tile and the same activation staging (LDS-DMA, swizzled, double buffer) as dense_gemm_kernel; the
// weights never touch LDS: word j of a lane in the tiled image holds exactly the 8 codes of its B fragment of MFMA step j (that is
// what the tile order is), so a wave fetches one dwordx4 per n-fragment and 128 k and unpacks it in registers (128+q / 1024+q by one
// AND-OR, code pairs (q_i, q_i+4): the A fragment is permuted to match on its way out of LDS).  The accumulators run over one
// quantisation group (a multiple of 64 k); at its end  y += s_g (acc - (OFF + z) xsum_g)  with the per-token group sums of x from a
// ones-MFMA (as the decode kernels) and (scale, zero point) from a panel staged in LDS once per workgroup: inside the K loop the only
// global traffic is the activation DMA and the weight ring -- a load that hipcc can see next to the invisible DMA makes it wait for
// everything (first version: scales and a pre-computed xsum array fetched at every group end, 2.1-2.4 us per K step, 214-306 TFLOP/s).
#define GG_NF 2
static_assert(GG_NF == 2, "the counted wait in gptq_gemm_kernel leaves GG_NF loads in flight");
template <int DT, int GPK>
__global__ void __launch_bounds__(256) gptq_gemm_kernel(const DenseArgs a) {
    extern __shared__ __attribute__((aligned(1024))) uint8_t smem[];   // [3 buffers][A 16 KiB]: the DMA runs two K steps ahead
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int m16 = lane & 15, kg = lane >> 4;
    const bool silu = a.epi == MI355_EPI_SILU_MUL;
    const int bm = blockIdx.x, bn = blockIdx.y;
    const int t0 = bm * DG_BM;
    const int n_cols = silu ? a.pair_offset : a.N;
    // 128 tokens x 64 columns per workgroup (SILU_MUL: 32 gate + the same 32 up columns): 4 m-fragments x 2 n-fragments per wave --
    // with 4 n-fragments the group accumulators, the running sums and the weight ring need more than 256 VGPRs
    const int c0 = bn * (silu ? 32 : 64);
    const int nkb = a.K >> 8, gs = a.group_size;
    // n-fragment j of this wave: first column (clamped to the matrix: columns beyond it are computed and never stored)
    int fcol[GG_NF];
#pragma unroll
    for (int j = 0; j < GG_NF; ++j) {
        int n = silu ? ((j ? a.pair_offset : 0) + c0 + wc * 16) : (c0 + wc * 32 + j * 16);
        const int lim = silu ? ((j ? a.N : a.pair_offset) - 16) : a.N - 16;
        fcol[j] = n > lim ? lim : n;
    }
    const uint32_t lds0 = __builtin_amdgcn_readfirstlane((uint32_t)(size_t)(__attribute__((address_space(3))) void*)smem);
    const uint8_t* xb = static_cast<const uint8_t*>(a.x);
    const int srow = tid >> 3, sslot = tid & 7;
    const uint8_t* asrc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = 32 * r + srow;
        int t = t0 + row;
        if (t > a.T - 1) t = a.T - 1;
        asrc[r] = xb + ((size_t)t * a.ldx) * 2 + (size_t)((sslot ^ (row & 7)) * 16);
    }
    // ---- (scale, zero point) panel of this workgroup's 64 columns, all groups: u32 = zero point (f32 bits >> 16, an exact small
    // integer) << 16 | scale bits; panel[g][wave-column wc][fragment j][m16]
    uint32_t* panel = reinterpret_cast<uint32_t*>(smem + 3 * 16384);
    {
        const int ng = a.K / gs;
        const uint16_t* sc16p = static_cast<const uint16_t*>(a.scales);
        for (int e = tid; e < ng * 64; e += 256) {
            const int g = e >> 6, lc = e & 63, pwc = lc >> 5, pj = (lc >> 4) & 1, pm = lc & 15;
            int n = silu ? ((pj ? a.pair_offset : 0) + c0 + pwc * 16) : (c0 + pwc * 32 + pj * 16);
            const int lim = silu ? ((pj ? a.N : a.pair_offset) - 16) : a.N - 16;
            const int col = (n > lim ? lim : n) + pm;
            const uint32_t sb = sc16p[(size_t)g * a.N + marlin_scale_pos(col, a.sperm)];
            const float z = zero_point(a, g, col);
            panel[e] = (__float_as_uint(z) & 0xFFFF0000u) | sb;
        }
    }
    auto stage = [&](int kt, int buf) {
        const uint32_t base = lds0 + (uint32_t)buf * 16384u + (uint32_t)wave * 1024u;
#pragma unroll
        for (int r = 0; r < 4; ++r) dgemm_dma<1>(asrc[r] + (size_t)kt * (DG_BK * 2), base + (uint32_t)r * 4096u);
    };
    // weights: one dwordx4 per n-fragment and 128 k: [tile][k-block][half][lane][4 words]
    const dg_u32x4* wbase[GG_NF];
#pragma unroll
    for (int j = 0; j < GG_NF; ++j) wbase[j] = reinterpret_cast<const dg_u32x4*>(a.w) + ((size_t)(fcol[j] >> 4) * nkb * 2) * 64 + lane;
    dg_u32x4 wq[2][GG_NF];
    auto wload = [&](int slot, int h128) {                               // h128 = index of the 128-k half (2 per k-block)
#pragma unroll
        for (int j = 0; j < GG_NF; ++j) wq[slot][j] = __builtin_nontemporal_load(wbase[j] + (size_t)h128 * 64);
    };
    constexpr uint32_t KOFF = (DT == MI355_DTYPE_BF16) ? 0x43004300u : 0x64006400u;
    constexpr float OFF = (DT == MI355_DTYPE_BF16) ? 128.f : 1024.f;
    uint32_t nib = 0x000F000Fu;
    asm volatile("" : "+v"(nib));
    constexpr uint32_t ONE2 = (DT == MI355_DTYPE_BF16) ? 0x3F803F80u : 0x3C003C00u;
    const uint4 ones = make_uint4(ONE2, ONE2, ONE2, ONE2);
    f32x4_t xs[4];                                                       // group sums of x per token (every column of a ones-MFMA holds them)
#pragma unroll
    for (int i = 0; i < 4; ++i) xs[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    f32x4_t acc[4][GG_NF], y[4][GG_NF];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < GG_NF; ++j) { acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f}; y[i][j] = acc[i][j]; }
    int arow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) arow[i] = wr * 64 + i * 16 + m16;
    const int nkt = a.K / DG_BK;
    const int nh = a.K >> 7;
    // A K step is 16 MFMAs per wave -- far less than a memory round trip -- so the activation tiles run TWO steps ahead through three
    // buffers and the end-of-step wait is counted: only the tile of the NEXT step has to have landed (measured with a two-buffer,
    // wait-for-everything loop: 2.4 us per K step, 214-306 TFLOP/s)
    stage(0, 0);
    if (nkt > 1) stage(1, 1);
    wload(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // One k-block (256 k = 4 K steps = 2 weight halves) per iteration of the run-time loop, its four steps unrolled: buffer, ring slot
    // and word indices are compile-time (a run-time slot index put the weight ring into scratch), and so is the position of the
    // per-group update (GPK = K steps per group: 1 / 2 / 4; 0 = whole k-blocks, the update sits behind a loop over the group's
    // k-blocks).  With the update under a run-time condition inside one flat loop hipcc parked y in the accumulation registers and
    // moved 200 registers per K step back and forth (v_accvgpr_read / write: as many issue cycles as the MFMAs).
    auto update = [&](const int g) {
#pragma unroll
        for (int j = 0; j < GG_NF; ++j) {
            const uint32_t pz = panel[g * 64 + wc * 32 + j * 16 + m16];
            const float sg = h2f<DT>((uint16_t)(pz & 0xFFFF));
            const float cg = -(OFF + __uint_as_float(pz & 0xFFFF0000u)) * sg;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int v = 0; v < 4; ++v) y[i][j][v] += sg * acc[i][j][v] + cg * xs[i][v];
                acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) xs[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    };
#define GG_STEP(S_) do { \
        constexpr int s_ = (S_); \
        const int kt = 4 * kb + s_; \
        const bool snext = kt + 2 < nkt; \
        if (snext) stage(kt + 2, (kt + 2) % 3); \
        constexpr bool wstep = !(s_ & 1); \
        const bool wnext = wstep && 2 * kb + (s_ >> 1) + 1 < nh; \
        if (wnext) wload(((s_ >> 1) & 1) ^ 1, 2 * kb + (s_ >> 1) + 1); \
        const uint8_t* A = smem + (size_t)(kt % 3) * 16384; \
        _Pragma("unroll") for (int s2 = 0; s2 < 2; ++s2) { \
            uint4 af[4]; \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) { \
                const uint4 v = *reinterpret_cast<const uint4*>(A + (size_t)arow[i] * 128 + (size_t)(((s2 * 4 + kg) ^ (arow[i] & 7)) * 16)); \
                af[i].x = __builtin_amdgcn_perm(v.z, v.x, 0x05040100u); \
                af[i].y = __builtin_amdgcn_perm(v.z, v.x, 0x07060302u); \
                af[i].z = __builtin_amdgcn_perm(v.w, v.y, 0x05040100u); \
                af[i].w = __builtin_amdgcn_perm(v.w, v.y, 0x07060302u); \
                xs[i] = mfma32<DT>(af[i], ones, xs[i]); \
            } \
            _Pragma("unroll") for (int j = 0; j < GG_NF; ++j) { \
                const dg_u32x4 wv = wq[(s_ >> 1) & 1][j]; \
                const uint32_t w = ((s_ & 1) * 2 + s2) == 0 ? wv.x : ((s_ & 1) * 2 + s2) == 1 ? wv.y : ((s_ & 1) * 2 + s2) == 2 ? wv.z : wv.w; \
                uint4 b; \
                b.x = (w & nib) | KOFF; b.y = ((w >> 4) & nib) | KOFF; b.z = ((w >> 8) & nib) | KOFF; b.w = ((w >> 12) & nib) | KOFF; \
                _Pragma("unroll") for (int i = 0; i < 4; ++i) acc[i][j] = mfma32<DT>(af[i], b, acc[i][j]); \
            } \
        } \
        /* the tile of step kt + 1 (DMA issued one step ago) must have landed; younger than it: the weight loads of the previous step \
           (odd steps) or of this one (even steps), and this step's DMA */ \
        { \
            const int younger = (snext ? 4 : 0) + ((wstep ? wnext : wprev) ? GG_NF : 0); \
            if (younger >= 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); \
            else if (younger == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); \
            else if (younger == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); \
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); \
        } \
        wprev = wnext; \
        __syncthreads(); \
        if constexpr (GPK == 1 || (GPK == 2 && (s_ & 1)) || (GPK == 4 && s_ == 3)) update(kt / (GPK ? GPK : 1)); \
    } while (0)
    bool wprev = false;
    if constexpr (GPK != 0) {
        for (int kb = 0; kb < nkb; ++kb) { GG_STEP(0); GG_STEP(1); GG_STEP(2); GG_STEP(3); }
    } else {
        const int kb_per_group = gs >> 8;
        for (int g = 0; g < nkb / kb_per_group; ++g) {
            for (int kbi = 0; kbi < kb_per_group; ++kbi) { const int kb = g * kb_per_group + kbi; GG_STEP(0); GG_STEP(1); GG_STEP(2); GG_STEP(3); }
            update(g);
        }
    }
#undef GG_STEP
    // ---- epilogue (as dense_gemm_kernel): the lane holds tokens 4 kg + v of every m-fragment for column m16 of every n-fragment
    const uint16_t* bias = static_cast<const uint16_t*>(a.bias);
    const uint16_t* resid = static_cast<const uint16_t*>(a.resid);
    uint16_t* out = static_cast<uint16_t*>(a.out);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int t = t0 + wr * 64 + i * 16 + 4 * kg + v;
            if (t >= a.T) continue;
            if (silu) {
                const int col = c0 + wc * 16 + m16;
                if (col < n_cols) {
                    float g = rnd<DT>(y[i][0][v]), u = rnd<DT>(y[i][1][v]);
                    if (bias) { g = rnd<DT>(g + h2f<DT>(bias[col])); u = rnd<DT>(u + h2f<DT>(bias[a.pair_offset + col])); }
                    out[(size_t)t * a.ldo + col] = f2h<DT>(rnd<DT>(rnd<DT>(g / (1.f + __expf(-g))) * u));
                }
            } else {
#pragma unroll
                for (int j = 0; j < GG_NF; ++j) {
                    const int col = c0 + wc * 32 + j * 16 + m16;
                    if (col >= n_cols) continue;
                    float o = rnd<DT>(y[i][j][v]);
                    if (bias) o = rnd<DT>(o + h2f<DT>(bias[col]));
                    if (a.epi == MI355_EPI_RESID) o = rnd<DT>(o + h2f<DT>(resid[(size_t)t * a.ldo + col]));
                    out[(size_t)t * a.ldo + col] = f2h<DT>(o);
                }
            }
        }
}
static int gptq_prompt_gemm(const DenseArgs& a, int dt, hipStream_t st) {
    const int gs = a.group_size;
    if (g_tune_gptq_gemm_off || (a.K & 255) || (a.N & 15) || a.T < 4 || gs < 64 || (gs % 64) || (a.K % gs)) return (int)hipErrorNotSupported;
    if ((a.ldx * 2) % 16 || ((uintptr_t)a.x & 15)) return (int)hipErrorNotSupported;
    if (a.epi == MI355_EPI_SILU_MUL && (a.pair_offset <= 0 || a.N != 2 * a.pair_offset || (a.pair_offset & 15))) return (int)hipErrorNotSupported;
    const int n_cols = a.epi == MI355_EPI_SILU_MUL ? a.pair_offset : a.N;
    const int cols_per_wg = a.epi == MI355_EPI_SILU_MUL ? 32 : 64;
    const dim3 grid((a.T + DG_BM - 1) / DG_BM, (n_cols + cols_per_wg - 1) / cols_per_wg);
    const size_t lds = 3 * 16384 + (size_t)(a.K / gs) * 64 * sizeof(uint32_t);
    if (lds > 152 * 1024) return (int)hipErrorNotSupported;
    const int gpk = gs == 64 ? 1 : gs == 128 ? 2 : gs == 256 ? 4 : 0;
    if (gpk == 0 && (gs % 256)) return (int)hipErrorNotSupported;
#define GG_GO(DT_, GPK_) do { \
        static bool attr_ = false; \
        if (!attr_) { (void)hipFuncSetAttribute((const void*)gptq_gemm_kernel<DT_, GPK_>, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024); attr_ = true; } \
        hipLaunchKernelGGL((gptq_gemm_kernel<DT_, GPK_>), grid, dim3(256), lds, st, a); } while (0)
#define GG_DT(DT_) do { if (gpk == 1) GG_GO(DT_, 1); else if (gpk == 2) GG_GO(DT_, 2); else if (gpk == 4) GG_GO(DT_, 4); else GG_GO(DT_, 0); } while (0)
    if (dt == MI355_DTYPE_BF16) GG_DT(MI355_DTYPE_BF16);
    else if (dt == MI355_DTYPE_F16) GG_DT(MI355_DTYPE_F16);
#undef GG_DT
#undef GG_GO
    else return (int)hipErrorNotSupported;
    return (int)hipGetLastError();
}

static int dense_prompt_gemm(const DenseArgs& a, int dt, hipStream_t st) {
    if (a.K % DG_BK || a.T < 1 || a.N < 1) return (int)hipErrorNotSupported;
    if ((a.ldx * 2) % 16 || (!a.wtiled && (a.ldw * 2) % 16)) return (int)hipErrorNotSupported;          // 16-byte DMA pieces
    if (a.epi == MI355_EPI_SILU_MUL && (a.pair_offset <= 0 || a.N != 2 * a.pair_offset)) return (int)hipErrorNotSupported;
    static Mi355DevOnce attr_done;
    if (!attr_done.done()) {
        (void)hipFuncSetAttribute((const void*)dense_gemm_kernel<MI355_DTYPE_BF16, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        (void)hipFuncSetAttribute((const void*)dense_gemm_kernel<MI355_DTYPE_F16, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        (void)hipFuncSetAttribute((const void*)dense_gemm_kernel<MI355_DTYPE_BF16, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        (void)hipFuncSetAttribute((const void*)dense_gemm_kernel<MI355_DTYPE_F16, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        attr_done.set();
    }
    const int n_cols = a.epi == MI355_EPI_SILU_MUL ? a.pair_offset : a.N;
    const int cols_per_wg = a.epi == MI355_EPI_SILU_MUL ? 64 : DG_BN;
    const int ncb = (n_cols + cols_per_wg - 1) / cols_per_wg;
    // the 256-token tile (one 8-wave workgroup per CU) when it still fills the chip twice
    const bool tall = g_tune_gemm_tall && ((g_tune_gemm_tall & 2) || (int64_t)((a.T + 255) / 256) * ncb >= 512);   // (bit 64 of key 30: wherever the GEMM runs -- tests)
    if (tall) {
        const dim3 grid((a.T + 255) / 256, ncb);
        if (dt == MI355_DTYPE_BF16) hipLaunchKernelGGL((dense_gemm_kernel<MI355_DTYPE_BF16, 4>), grid, dim3(512), 96 * 1024, st, a);
        else if (dt == MI355_DTYPE_F16) hipLaunchKernelGGL((dense_gemm_kernel<MI355_DTYPE_F16, 4>), grid, dim3(512), 96 * 1024, st, a);
        else return (int)hipErrorNotSupported;
        return (int)hipGetLastError();
    }
    const dim3 grid((a.T + DG_BM - 1) / DG_BM, ncb);
    if (dt == MI355_DTYPE_BF16) hipLaunchKernelGGL((dense_gemm_kernel<MI355_DTYPE_BF16, 2>), grid, dim3(256), 64 * 1024, st, a);
    else if (dt == MI355_DTYPE_F16) hipLaunchKernelGGL((dense_gemm_kernel<MI355_DTYPE_F16, 2>), grid, dim3(256), 64 * 1024, st, a);
    else return (int)hipErrorNotSupported;
    return (int)hipGetLastError();
}

// SILU_MUL with more than 32 tokens, or any T > 64: run in token chunks
static int dense_run(DenseArgs a, int wtype, int dt, hipStream_t st) {
    if (wtype == DW_DENSE && a.T >= g_tune_gemm_min_t) {
        const int rc = dense_prompt_gemm(a, dt, st);
        if (rc != (int)hipErrorNotSupported) return rc;                    // odd strides / K: keep streaming in token chunks
    }
    if (wtype == DW_GPTQ4T && a.T >= g_tune_gemm_min_t) {
        const int rc = gptq_prompt_gemm(a, dt, st);
        if (rc != (int)hipErrorNotSupported) return rc;                    // groups of 32, odd strides: token chunks of the decode kernel
    }
    const int chunk = a.epi == MI355_EPI_SILU_MUL ? 32 : 64;
    const int T = a.T;
    if (T < 1) return -2;
    for (int t0 = 0; t0 < T; t0 += chunk) {
        DenseArgs c = a;
        c.T = T - t0 < chunk ? T - t0 : chunk;
        c.x = static_cast<const uint16_t*>(a.x) + (size_t)t0 * a.ldx;
        c.out = static_cast<uint16_t*>(a.out) + (size_t)t0 * a.ldo;
        if (a.resid) c.resid = static_cast<const uint16_t*>(a.resid) + (size_t)t0 * a.ldo;
        const int rc = dense_dispatch(c, wtype, dt, st);
        if (rc) return rc;
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------ K15 repack
// 16-bit [N][ld_in] row-major -> the tiled image (dense_tile_index): one thread per 16-byte chunk of the output
__global__ void __launch_bounds__(256) dense_tile_kernel(const uint16_t* __restrict__ in, uint16_t* __restrict__ out, int N, int K, int64_t ld_in) {
    const size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x;                 // chunk index = (((tile * nkb + kb) * 8 + j) * 64 + lane)
    const int nkb = K >> 8;
    if (o >= (size_t)(N >> 4) * nkb * 512) return;
    const int lane = (int)(o & 63), j = (int)((o >> 6) & 7);
    const size_t tk = o >> 9;
    const int kb = (int)(tk % nkb), tile = (int)(tk / nkb);
    const int n = tile * 16 + (lane & 15), k = kb * 256 + j * 32 + (lane >> 4) * 8;
    *reinterpret_cast<uint4*>(out + o * 8) = *reinterpret_cast<const uint4*>(in + (size_t)n * ld_in + k);
}
// checkpoint layout [K/8][n_in] (columns [0, n_in)) -> tiles tile0 .. of the tiled image (gptq_tile_index); K % 256 == 0, n_in % 16 == 0
__global__ void gptq_tile_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, int K, int n_in, int tile0) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)(K / 8) * n_in) return;
    const int n = (int)(idx % n_in), kr = (int)(idx / n_in);
    out[gptq_tile_index(kr, n + 16 * tile0, K >> 8)] = in[idx];
}
// the inverse (tests / the exllama arm of a model that changes its mind): tiled image -> [K/8][N]
__global__ void gptq_untile_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, int K, int N) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)(K / 8) * N) return;
    out[idx] = in[gptq_tile_index((int)(idx / N), (int)(idx % N), K >> 8)];
}
static inline bool gptq_tileable(int k, int n) { return k > 0 && n > 0 && !(k & 255) && !(n & 15); }

__global__ void awq_repack_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, int K, int NP, int tiled) {
    // in [K][NP] : nibble i of in[k][c] = code(k, 8c + {0,2,4,6,1,3,5,7}[i])   [EXT: AutoAWQ order_map]
    // out [K/8][8 NP] : nibble i of out[kr][n] = code(8kr + i, n), stored at gptq_tile_index(kr, n) when `tiled`
    const int N = NP * 8;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)(K / 8) * N) return;
    const int n = (int)(idx % N), kr = (int)(idx / N);
    const int u = n & 7;
    const int pos = ((u & 1) << 2) | (u >> 1);
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) o |= ((in[(size_t)(8 * kr + i) * NP + (n >> 3)] >> (4 * pos)) & 0xFu) << (4 * i);
    out[tiled ? gptq_tile_index(kr, n, K >> 8) : idx] = o;
}

// ------------------------------------------------------------------------------------------------ K14 exllama
// act-order GPTQ: every k has its own group g_idx[k], so the scale cannot be factored out of an MFMA chain; the
// weights are dequantised to f16 (w = s * (q - (z+1)) rounded to f16, as a half-precision dequant kernel does)
// and contracted on MFMA.  One workgroup per 16-column tile, its 4 waves split K.
template <int MT, int BITS>
__global__ void __launch_bounds__(256) exllama_kernel(const uint16_t* __restrict__ x, const uint32_t* __restrict__ qw,
                                                      const uint32_t* __restrict__ qz, const uint16_t* __restrict__ sc,
                                                      const int32_t* __restrict__ g_idx, uint16_t* __restrict__ out,
                                                      int T, int N, int K, int group_size) {
    __shared__ float red[4][MT][16][16];
    constexpr int PF = 32 / BITS;                                   // codes per u32: 8 (4-bit) or 4 (8-bit, linear.rs:215-217)
    constexpr uint32_t QM = (1u << BITS) - 1u;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r16 = lane & 15, kg = lane >> 4;
    const int col = blockIdx.x * 16 + r16;
    f32x4_t y[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) y[mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    for (int k32 = wave; k32 < K / 32; k32 += 4) {
        const int k0 = k32 * 32 + 8 * kg;
        uint32_t w[8 / PF];                                         // the 8 consecutive k of this lane: 1 word (4-bit) / 2 words (8-bit)
#pragma unroll
        for (int u = 0; u < 8 / PF; ++u) w[u] = qw[(size_t)(k0 / PF + u) * N + col];
        f16x8_t bf;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int g = g_idx ? g_idx[k0 + i] : (k0 + i) / group_size;
            const float s = f16_bits_to_f32(sc[(size_t)g * N + col]);
            const int z = (int)((qz[(size_t)g * (N / PF) + col / PF] >> (BITS * (col % PF))) & QM) + 1;
            bf[i] = (_Float16)(s * (float)((int)((w[i / PF] >> (BITS * (i % PF))) & QM) - z));
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int m = min(mt * 16 + r16, T - 1);
            const f16x8_t af = *reinterpret_cast<const f16x8_t*>(x + (size_t)m * K + k0);
            y[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf, y[mt], 0, 0, 0);
        }
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int v = 0; v < 4; ++v) red[wave][mt][4 * kg + v][r16] = y[mt][v];
    __syncthreads();
    for (int idx = threadIdx.x; idx < MT * 256; idx += 256) {
        const int rr = idx & 15, m = idx >> 4;
        if (m >= T) continue;
        const float s = red[0][m >> 4][m & 15][rr] + red[1][m >> 4][m & 15][rr] + red[2][m >> 4][m & 15][rr] + red[3][m >> 4][m & 15][rr];
        out[(size_t)m * N + blockIdx.x * 16 + rr] = f32_to_f16_bits(s);
    }
}

// ------------------------------------------------------------------------------------------------ Marlin-format checkpoints
// `checkpoint_format == "marlin"` (linear.rs:222-239,279-290): the file holds B [k/16, 2n] u32 already in the Marlin
// tile order (IST-DASLab marlin `Layer.pack` [EXT]): codes (k, n) -> 16x16 tiles, tile-row major; every 1024
// consecutive values of a tile row (4 tiles) permuted by `_perm`, then 8 values per word with the nibble
// interleave [0,2,4,6,1,3,5,7].  The reference hands B to marlin_4bit_* without a repack; this path's mat-mul streams
// the GPTQ layout qweight[k/8][n], so a Marlin-format B is un-permuted ONCE at load time.
//   perm1(i)[4*blk + e] = 16*row_e(i) + i/4 + 8*blk,  row_e = {2(i%4), 2(i%4)+1, 2(i%4+4), 2(i%4+4)+1}
//   perm = concat_i concat_j (perm1(i) + 256 j), then within every 8: [0,2,4,6,1,3,5,7]
__device__ __forceinline__ int marlin_perm_at(int j) {              // j in [0,1024): source index inside the 1024 chunk
    const int e8 = j & 7, grp = j >> 3;                             // 8-group; interleave picks element {0,2,4,6,1,3,5,7}[e8]
    const int src8 = (e8 < 4) ? 2 * e8 : 2 * (e8 - 4) + 1;
    const int q = grp * 8 + src8;                                   // index into the un-interleaved perm
    const int i = q >> 5, jj = (q >> 3) & 3, t = q & 7;             // 32 entries per i: 4 (j) x 8 (perm1)
    const int blk = t >> 2, e = t & 3;
    const int r = (e < 2) ? 2 * (i & 3) + e : 2 * ((i & 3) + 4) + (e - 2);
    return 16 * r + (i >> 2) + 8 * blk + 256 * jj;
}
__global__ void __launch_bounds__(256) marlin_to_gptq_kernel(const uint32_t* __restrict__ B, uint32_t* __restrict__ out, int K, int N, int tiled) {
    __shared__ uint16_t inv[1024];
    for (int j = threadIdx.x; j < 1024; j += 256) inv[marlin_perm_at(j)] = (uint16_t)j;
    __syncthreads();
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)(K / 8) * N) return;
    const int n = (int)(idx % N), kr = (int)(idx / N);
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int k = 8 * kr + i;
        const int src = (n >> 4) * 256 + (k & 15) * 16 + (n & 15);  // position inside tile row k/16 before the permutation
        const int col = (src & ~1023) + inv[src & 1023];
        o |= ((B[(size_t)(k >> 4) * (2 * N) + (col >> 3)] >> (4 * (col & 7))) & 0xFu) << (4 * i);
    }
    out[tiled ? gptq_tile_index(kr, n, K >> 8) : idx] = o;
}

// ------------------------------------------------------------------------------------------------ C ABI
static int marlin_common(const void* in, const int32_t* qweight, const void* scales, const void* zeros, void* out,
                         int m, int k, int n, int group_size, int dt, int awq, int64_t stream) {
    DenseArgs a{};
    a.w = qweight; a.scales = scales;
    a.group_size = (group_size <= 0 || group_size > k) ? k : group_size;
    a.sperm = a.group_size < k ? SP_GROUPED : SP_SINGLE;
    a.qzeros = awq ? static_cast<const uint32_t*>(zeros) : nullptr;
    a.zmode = awq ? MI355_ZERO_AWQ_MARLIN : MI355_ZERO_SYM8;
    a.x = in; a.ldx = k; a.T = m; a.N = n; a.K = k;
    a.out = out; a.ldo = n; a.epi = MI355_EPI_STORE;
    return dense_run(a, DW_GPTQ4T, dt, (hipStream_t)stream);          // the image gptq_repack / awq_repack / mi355_marlin_format_repack made
}

template <int BITS>
static void exllama_launch(const uint16_t* x, const uint32_t* qw, const uint32_t* qz, const uint16_t* sc, const int32_t* g_idx,
                           uint16_t* o, int T, int n, int k, hipStream_t st) {
    // without g_idx the whole of k is one group: the op does not pass group_size, and the reference always loads g_idx
    // for gptq checkpoints (linear.rs:298-314)
    if (T <= 16) hipLaunchKernelGGL((exllama_kernel<1, BITS>), dim3(n / 16), dim3(256), 0, st, x, qw, qz, sc, g_idx, o, T, n, k, k);
    else hipLaunchKernelGGL((exllama_kernel<2, BITS>), dim3(n / 16), dim3(256), 0, st, x, qw, qz, sc, g_idx, o, T, n, k, k);
}
extern "C" {

// The reference's FFI symbols return nothing.  A call this library cannot serve must not leave the caller with an
// uninitialised output that looks like data: the failure is recorded (mi355_last_error) and the output is filled with
// 0xFFFF (NaN in f16 and bf16) whenever its pointer and shape are usable.
static void ffi_fail(int code, void* out, int64_t m, int64_t n, int64_t stream) {
    mi355_note_error(code ? code : (int)hipErrorInvalidValue);
    if (out && m > 0 && n > 0) (void)hipMemsetAsync(out, 0xFF, (size_t)m * n * 2, (hipStream_t)stream);
}
static void marlin_ffi(const void* in, const int32_t* qweight, const void* scales, const void* zeros, void* out, int m, int k, int n,
                       int group_size, int dt, int awq, int64_t stream) {
    if (m == 0) return;
    if (!in || !qweight || !scales || !out || m < 0 || k <= 0 || n <= 0 || (k & 255) || (n & 15) || (awq && !zeros)) {
        ffi_fail((int)hipErrorInvalidValue, out, m, n, stream);
        return;
    }
    const int rc = marlin_common(in, qweight, scales, zeros, out, m, k, n, group_size, dt, awq, stream);
    if (rc) ffi_fail(rc, out, m, n, stream);
}
void marlin_4bit_f16(const void* in, const int32_t* qweight, const void* scales, const void* zeros, const void* g_idx,
                     void* out, int32_t m, int32_t k, int32_t n, const void* workspace, int32_t group_size, int64_t stream) {
    (void)g_idx; (void)workspace;
    marlin_ffi(in, qweight, scales, zeros, out, m, k, n, group_size, MI355_DTYPE_F16, 0, stream);
}
void marlin_4bit_bf16(const void* in, const int32_t* qweight, const void* scales, const void* zeros, const void* g_idx,
                      void* out, int32_t m, int32_t k, int32_t n, const void* workspace, int32_t group_size, int64_t stream) {
    (void)g_idx; (void)workspace;
    marlin_ffi(in, qweight, scales, zeros, out, m, k, n, group_size, MI355_DTYPE_BF16, 0, stream);
}
void marlin_awq_4bit_f16(const void* in, const int32_t* qweight, const void* scales, const void* zeros, const void* g_idx,
                         void* out, int32_t m, int32_t k, int32_t n, const void* workspace, int32_t group_size, int64_t stream) {
    (void)g_idx; (void)workspace;
    marlin_ffi(in, qweight, scales, zeros, out, m, k, n, group_size, MI355_DTYPE_F16, 1, stream);
}
void marlin_awq_4bit_bf16(const void* in, const int32_t* qweight, const void* scales, const void* zeros, const void* g_idx,
                          void* out, int32_t m, int32_t k, int32_t n, const void* workspace, int32_t group_size, int64_t stream) {
    (void)g_idx; (void)workspace;
    marlin_ffi(in, qweight, scales, zeros, out, m, k, n, group_size, MI355_DTYPE_BF16, 1, stream);
}

void gemm_half_q_half_alt(const void* a, const uint32_t* b_q_weight, const uint32_t* b_gptq_qzeros, const void* b_gptq_scales,
                          const int32_t* b_g_idx, void* c, int32_t m, int32_t n, int32_t k, int32_t bit, int64_t stream) {
    if (m == 0) return;
    // the reference admits 4 | 8 bits on this arm (linear.rs:215-217) and forwards `bits` (gptq.rs:181-194)
    if ((bit != 4 && bit != 8) || !a || !b_q_weight || !b_gptq_qzeros || !b_gptq_scales || !c || m < 0 || n <= 0 || k <= 0 ||
        (n & 15) || (k & 31)) {
        ffi_fail((int)hipErrorInvalidValue, c, m, n, stream);
        return;
    }
    hipStream_t st = (hipStream_t)stream;
    for (int t0 = 0; t0 < m; t0 += 32) {
        const int T = m - t0 < 32 ? m - t0 : 32;
        const uint16_t* x = static_cast<const uint16_t*>(a) + (size_t)t0 * k;
        uint16_t* o = static_cast<uint16_t*>(c) + (size_t)t0 * n;
        if (bit == 4) exllama_launch<4>(x, b_q_weight, b_gptq_qzeros, static_cast<const uint16_t*>(b_gptq_scales), b_g_idx, o, T, n, k, st);
        else exllama_launch<8>(x, b_q_weight, b_gptq_qzeros, static_cast<const uint16_t*>(b_gptq_scales), b_g_idx, o, T, n, k, st);
    }
    const int rc = (int)hipGetLastError();
    if (rc) ffi_fail(rc, c, m, n, stream);
}

/* gptq_repack (marlin gptq_marlin_repack.cu's slot, linear.rs:845-853): checkpoint qweight [k/8][n] -> the weight image the
 * marlin_* entry points stream.  Every shape those entry points accept (k % 256 == 0, n % 16 == 0) gets the 16 x 256 tile order
 * (gptq_tile_index); any other shape is copied as is (marlin_* refuses it loudly either way).  in != out. */
void gptq_repack(const void* in, void* out, int32_t k_packed, int32_t n, int64_t stream) {
    if (!in || !out || k_packed <= 0 || n <= 0 || in == out) { mi355_note_error((int)hipErrorInvalidValue); return; }
    if (gptq_tileable(k_packed * 8, n)) {
        const size_t total = (size_t)k_packed * n;
        hipLaunchKernelGGL(gptq_tile_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                           static_cast<const uint32_t*>(in), static_cast<uint32_t*>(out), k_packed * 8, n, 0);
        const int rc = (int)hipGetLastError();
        if (rc) mi355_note_error(rc);
        return;
    }
    const hipError_t e = hipMemcpyAsync(out, in, (size_t)k_packed * n * sizeof(uint32_t), hipMemcpyDeviceToDevice, (hipStream_t)stream);
    if (e != hipSuccess) mi355_note_error((int)e);
}
/* the tile order on its own: columns [0, n_in) of a checkpoint-layout tensor -> tiles tile0 .. of `out` (a tensor of any width
 * >= 16 (tile0 + n_in / 16)): the host layer packs gate_proj and up_proj side by side this way.  And its inverse. */
int mi355_gptq_tile_repack(const void* in, void* out, int32_t k, int32_t n_in, int32_t tile0, int64_t stream) {
    if (!in || !out || in == out || !gptq_tileable(k, n_in) || tile0 < 0) return (int)hipErrorInvalidValue;
    const size_t total = (size_t)(k / 8) * n_in;
    hipLaunchKernelGGL(gptq_tile_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       static_cast<const uint32_t*>(in), static_cast<uint32_t*>(out), k, n_in, tile0);
    return (int)hipGetLastError();
}
int mi355_gptq_tile_unpack(const void* in, void* out, int32_t k, int32_t n, int64_t stream) {
    if (!in || !out || in == out || !gptq_tileable(k, n)) return (int)hipErrorInvalidValue;
    const size_t total = (size_t)(k / 8) * n;
    hipLaunchKernelGGL(gptq_untile_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       static_cast<const uint32_t*>(in), static_cast<uint32_t*>(out), k, n);
    return (int)hipGetLastError();
}
void awq_repack(const void* in, void* out, int32_t k, int32_t n_packed, int32_t bits, int64_t stream) {
    if (bits != 4 || (k & 7) || !in || !out || k <= 0 || n_packed <= 0) {
        // the output is a weight image, not activations: poison it as NaN-scaled codes cannot be expressed -- record and leave
        mi355_note_error((int)hipErrorInvalidValue);
        return;
    }
    const size_t total = (size_t)(k / 8) * n_packed * 8;
    hipLaunchKernelGGL(awq_repack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       static_cast<const uint32_t*>(in), static_cast<uint32_t*>(out), k, n_packed, gptq_tileable(k, n_packed * 8) ? 1 : 0);
    const int rc = (int)hipGetLastError();
    if (rc) mi355_note_error(rc);
}

/* Marlin-FORMAT checkpoint weight B [k/16, 2n] u32 -> the layout the marlin_* entry points stream ([k/8][n] u32).  The
 * integration calls this once per tensor on the `marlin_format` branch of qlinear (linear.rs:279-290), where the reference
 * passes B through untouched. */
int mi355_marlin_format_repack(const void* in_B, void* out, int32_t k, int32_t n, int64_t stream) {
    if (!in_B || !out || k <= 0 || n <= 0 || (k & 15) || (n & 63)) return (int)hipErrorInvalidValue;
    const size_t total = (size_t)(k / 8) * n;
    hipLaunchKernelGGL(marlin_to_gptq_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       static_cast<const uint32_t*>(in_B), static_cast<uint32_t*>(out), k, n, gptq_tileable(k, n) ? 1 : 0);
    return (int)hipGetLastError();
}
/* host statement of the tile order (unit-tested on CPU against oracle/gptq.py's gptq_tile); -1 for a shape that is not tiled */
int64_t mi355_gptq_tile_index(int32_t kr, int32_t n, int32_t k, int32_t n_total) {
    if (!gptq_tileable(k, n_total) || kr < 0 || kr >= k / 8 || n < 0 || n >= n_total) return -1;
    return (int64_t)gptq_tile_index(kr, n, k >> 8);
}
/* host statement of the same permutation (unit-tested on CPU against the numpy restatement of marlin's `_get_perms`) */
int32_t mi355_marlin_weight_perm(int32_t j) {
    if (j < 0 || j >= 1024) return -1;
    const int e8 = j & 7, grp = j >> 3;
    const int src8 = (e8 < 4) ? 2 * e8 : 2 * (e8 - 4) + 1;
    const int q = grp * 8 + src8;
    const int i = q >> 5, jj = (q >> 3) & 3, t = q & 7;
    const int blk = t >> 2, e = t & 3;
    const int r = (e < 2) ? 2 * (i & 3) + e : 2 * ((i & 3) + 4) + (e - 2);
    return 16 * r + (i >> 2) + 8 * blk + 256 * jj;
}

/* host-side index arithmetic of the Marlin permutations (unit-tested on CPU against the reference's Python) */
int32_t mi355_marlin_scale_pos(int32_t n, int32_t grouped) { return marlin_scale_pos(n, grouped ? SP_GROUPED : SP_SINGLE); }
int32_t mi355_marlin_zero_pos(int32_t n) {
    const int p1 = marlin_scale_pos(n, SP_GROUPED);
    const int u = p1 & 7;
    return (p1 & ~7) + (((u & 1) << 2) | (u >> 1));
}

/* 16-bit weights [n][k] (row stride ld_in elements) -> the 16-row x 256-k tile image mi355_linear_tiled streams; in != out */
int mi355_dense_tile_repack(const void* in, void* out, int32_t n, int32_t k, int64_t ld_in, int64_t stream) {
    if (!in || !out || in == out || n <= 0 || k <= 0 || (n & 15) || (k & 255) || ld_in < k || (ld_in & 7) || ((uintptr_t)in & 15) || ((uintptr_t)out & 15))
        return (int)hipErrorInvalidValue;
    const size_t chunks = (size_t)(n >> 4) * (k >> 8) * 512;
    hipLaunchKernelGGL(dense_tile_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       static_cast<const uint16_t*>(in), static_cast<uint16_t*>(out), n, k, ld_in);
    return (int)hipGetLastError();
}
/* host statement of the tile order (element index of w[n][k]; -1 for a shape that is not tiled) */
int64_t mi355_dense_tile_index(int32_t n_row, int32_t k_col, int32_t n, int32_t k) {
    if (n <= 0 || k <= 0 || (n & 15) || (k & 255) || n_row < 0 || n_row >= n || k_col < 0 || k_col >= k) return -1;
    return (int64_t)dense_tile_index(n_row, k_col, k >> 8);
}
static int linear_impl(int tiled, void* out, const void* x, const void* w, const void* bias, const void* residual, int32_t num_tokens,
                       int32_t n, int32_t k, int32_t dtype, int32_t epilogue, int64_t stream) {
    if (tiled && ((n & 15) || (k & 255))) return -2;
    DenseArgs a{};
    a.w = w; a.ldw = k; a.wtiled = tiled;
    a.x = x; a.ldx = k; a.T = num_tokens; a.K = k; a.N = n;
    a.bias = bias; a.resid = residual; a.epi = epilogue; a.out = out;
    if (epilogue == MI355_EPI_SILU_MUL) { a.pair_offset = n / 2; a.ldo = n / 2; } else a.ldo = n;
    if (epilogue == MI355_EPI_RESID && !residual) return -2;
    return dense_run(a, DW_DENSE, dtype, (hipStream_t)stream);
}
/* `Linear::forward` over the tiled weight image (mi355_dense_tile_repack); everything else as mi355_linear */
int mi355_linear_tiled(void* out, const void* x, const void* w_tiled, const void* bias, const void* residual, int32_t num_tokens,
                       int32_t n, int32_t k, int32_t dtype, int32_t epilogue, int64_t stream) {
    return linear_impl(1, out, x, w_tiled, bias, residual, num_tokens, n, k, dtype, epilogue, stream);
}
int mi355_linear(void* out, const void* x, const void* w, const void* bias, const void* residual, int32_t num_tokens,
                 int32_t n, int32_t k, int32_t dtype, int32_t epilogue, int64_t stream) {
    DenseArgs a{};
    a.w = w; a.ldw = k; a.x = x; a.ldx = k; a.T = num_tokens; a.K = k; a.N = n;
    a.bias = bias; a.resid = residual; a.epi = epilogue; a.out = out;
    if (epilogue == MI355_EPI_SILU_MUL) { a.pair_offset = n / 2; a.ldo = n / 2; } else a.ldo = n;
    if (epilogue == MI355_EPI_RESID && !residual) return -2;
    return dense_run(a, DW_DENSE, dtype, (hipStream_t)stream);
}

static int gptq_linear_impl(int wtype, void* out, const void* x, const void* qweight, const void* scales, const void* qzeros,
                            int32_t zero_mode, int32_t scales_permuted, const void* bias, const void* residual,
                            int32_t num_tokens, int32_t n, int32_t k, int32_t group_size, int32_t dtype, int32_t epilogue,
                            int64_t stream) {
    DenseArgs a{};
    a.w = qweight; a.scales = scales; a.qzeros = static_cast<const uint32_t*>(qzeros); a.zmode = zero_mode;
    a.group_size = (group_size <= 0 || group_size > k) ? k : group_size;
    a.sperm = !scales_permuted ? SP_NONE : (a.group_size < k ? SP_GROUPED : SP_SINGLE);
    a.x = x; a.ldx = k; a.T = num_tokens; a.K = k; a.N = n;
    a.bias = bias; a.resid = residual; a.epi = epilogue; a.out = out;
    if (epilogue == MI355_EPI_SILU_MUL) { a.pair_offset = n / 2; a.ldo = n / 2; } else a.ldo = n;
    if (epilogue == MI355_EPI_RESID && !residual) return -2;
    return dense_run(a, wtype, dtype, (hipStream_t)stream);
}
int mi355_gptq_linear(void* out, const void* x, const void* qweight, const void* scales, const void* qzeros,
                      int32_t zero_mode, int32_t scales_permuted, const void* bias, const void* residual,
                      int32_t num_tokens, int32_t n, int32_t k, int32_t group_size, int32_t dtype, int32_t epilogue,
                      int64_t stream) {
    return gptq_linear_impl(DW_GPTQ4, out, x, qweight, scales, qzeros, zero_mode, scales_permuted, bias, residual, num_tokens, n, k,
                            group_size, dtype, epilogue, stream);
}
/* host layer only: the tiled sym-8 op on the 1..4-token kernel, with its two extras: RmsNorm(norm_w, eps) applied to x on the
 * way in (inv from `ss_in`, the per-tile sums of squares the producer of x left behind) and `ss_out`, the same sums of this
 * launch's own output rows for the next consumer.  -4 when the shape is not served by that kernel: the caller then norms
 * separately, calls mi355_gptq_linear_tiled, and treats ss_out as not written. */
int mi355_internal_gptq_small_linear(void* out, const void* x, const void* qweight_tiled, const void* scales, const void* bias,
                                     const void* residual, const void* norm_w, float norm_eps, const float* ss_in, float* ss_out,
                                     int32_t num_tokens, int32_t n, int32_t k, int32_t group_size, int32_t dtype, int32_t epilogue,
                                     int64_t stream) {
    if (num_tokens < 1 || num_tokens > 4 || (epilogue == MI355_EPI_RESID && !residual)) return -4;
    DenseArgs a{};
    a.w = qweight_tiled; a.scales = scales; a.zmode = MI355_ZERO_SYM8;
    a.group_size = (group_size <= 0 || group_size > k) ? k : group_size;
    a.sperm = SP_NONE;
    a.x = x; a.ldx = k; a.T = num_tokens; a.K = k; a.N = n;
    a.bias = bias; a.resid = residual; a.epi = epilogue; a.out = out;
    a.norm_w = norm_w; a.norm_eps = norm_eps; a.ss_in = ss_in; a.ss_out = ss_out;
    if (epilogue == MI355_EPI_SILU_MUL) { a.pair_offset = n / 2; a.ldo = n / 2; } else a.ldo = n;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MI355_DTYPE_BF16) return dense_small_launch<MI355_DTYPE_BF16, DW_GPTQ4T>(a, st);
    if (dtype == MI355_DTYPE_F16) return dense_small_launch<MI355_DTYPE_F16, DW_GPTQ4T>(a, st);
    return -4;
}
/* the same op over the TILED weight image (gptq_repack / mi355_gptq_tile_repack); scales and zero points as above */
int mi355_gptq_linear_tiled(void* out, const void* x, const void* qweight_tiled, const void* scales, const void* qzeros,
                            int32_t zero_mode, int32_t scales_permuted, const void* bias, const void* residual,
                            int32_t num_tokens, int32_t n, int32_t k, int32_t group_size, int32_t dtype, int32_t epilogue,
                            int64_t stream) {
    return gptq_linear_impl(DW_GPTQ4T, out, x, qweight_tiled, scales, qzeros, zero_mode, scales_permuted, bias, residual, num_tokens,
                            n, k, group_size, dtype, epilogue, stream);
}

/* q, k, v projections (plain store epilogue, optional bias) in one launch; returns -4 when the shapes do not qualify
 * (the caller then issues three mi355_linear / mi355_gptq_linear calls).  norm_w != NULL: x is the residual stream and the
 * launch applies RmsNorm(norm_w, eps) on the way in, inv from the producer's sums of squares `ss_in` (1..4 tokens of 4-bit
 * weights only; -4 otherwise and the caller norms separately).  rope != NULL (a DenseRope): the launch also rotates q and k
 * and writes k and v into the layer's bf16 cache -- the work of mi355_internal_rope_cache -- in its epilogue (same kernel only;
 * -4 otherwise and the caller keeps the separate launch).  Internal to the host layer (dense_model.cpp). */
int mi355_internal_linear3(void* const* outs, const void* x, const void* const* ws, const void* const* scales,
                           const void* const* biases, const int32_t* ns, int32_t num_tokens, int32_t k, int32_t group_size,
                           int32_t is_gptq, int32_t dtype, const void* norm_w, float norm_eps, const float* ss_in, const void* rope,
                           int64_t stream) {
    if (num_tokens < 1 || num_tokens > 64 || (k & 255)) return -4;
    const DenseRope* rp = static_cast<const DenseRope*>(rope);
    DenseArgs a[3];
    for (int i = 0; i < 3; ++i) {
        if (!ws[i] || !outs[i] || ns[i] <= 0 || (ns[i] & 15)) return -4;
        a[i] = DenseArgs{};
        a[i].w = ws[i]; a[i].ldw = k; a[i].wtiled = is_gptq == 3 ? 1 : 0; a[i].x = x; a[i].ldx = k; a[i].T = num_tokens; a[i].K = k; a[i].N = ns[i];
        a[i].bias = biases ? biases[i] : nullptr; a[i].epi = MI355_EPI_STORE; a[i].out = outs[i]; a[i].ldo = ns[i];
        a[i].norm_w = norm_w; a[i].norm_eps = norm_eps; a[i].ss_in = ss_in;
        if (is_gptq == 1 || is_gptq == 2) {
            a[i].scales = scales[i]; a[i].qzeros = nullptr; a[i].zmode = MI355_ZERO_SYM8;
            a[i].group_size = (group_size <= 0 || group_size > k) ? k : group_size;
            a[i].sperm = SP_NONE;
        }
    }
    hipStream_t st = (hipStream_t)stream;
    if (dtype == MI355_DTYPE_BF16)
        return is_gptq == 2 ? dense3_launch_dt<MI355_DTYPE_BF16, DW_GPTQ4T>(a, rp, st)
               : is_gptq == 1 ? dense3_launch_dt<MI355_DTYPE_BF16, DW_GPTQ4>(a, rp, st) : dense3_launch_dt<MI355_DTYPE_BF16, DW_DENSE>(a, rp, st);
    if (dtype == MI355_DTYPE_F16)
        return is_gptq == 2 ? dense3_launch_dt<MI355_DTYPE_F16, DW_GPTQ4T>(a, rp, st)
               : is_gptq == 1 ? dense3_launch_dt<MI355_DTYPE_F16, DW_GPTQ4>(a, rp, st) : dense3_launch_dt<MI355_DTYPE_F16, DW_DENSE>(a, rp, st);
    return -4;
}

}  // extern "C"
