// GGUF k-quant (Q4_K / Q6_K) dequant-matmul for gfx950 (MI355X): K11 `QMatMul::forward`.
//
// Reference boundary: candle `QMatMul::forward(&x_f32)` as called from
//   src/openai/models/layers/attention.rs:920-922,1004 (wq/wk/wv/wo),
//   src/openai/models/quantized_llama.rs:33-37 (w1/w3/w2), src/openai/distributed.rs:1633 (lm_head);
//   x is f32 [T,K], W is [N, K/256] super-blocks, y is f32 [T,N].
//
// Decode (T <= 8 per M-tile) is a pure HBM stream of the weights: 144 B (Q4_K) / 210 B (Q6_K) per 256
// weights, each byte read exactly once.  Design:
//   * weights are re-tiled once at load time (mi355_qweight_repack) into 16-row x 256-k tiles whose
//     quant payload is lane-linear: every wave load is one contiguous 1-KiB `global_load_dwordx4`;
//     a row tile's k-blocks are contiguous, so a workgroup streams one contiguous span of HBM;
//   * the int4/int6 codes are unpacked IN REGISTERS to bf16 (128+q is exact in bf16: 0x4300|q) and fed
//     straight to MFMA (v_mfma_f32_16x16x32_bf16, one MFMA = one 32-weight sub-block so the per-sub-block
//     scale is applied to the fp32 MFMA result; Q6_K uses the K=16 MFMA for its 16-weight sub-blocks);
//   * activations stay fp32-accurate: x is split x = hi + lo (two bf16), hi rows and lo rows ride in the
//     same MFMA as extra M rows (rows 0-7 hi, 8-15 lo) and are summed after scaling (error ~2^-17);
//   * the "+128" code offset and the Q4_K minimum are folded into one term  c_j * sum_k(x)  per sub-block;
//   * x (optionally RMS-normalised on the fly: fused K8) is staged once per workgroup into LDS in
//     MFMA-fragment order while the first weight loads are already in flight; waves of a workgroup split
//     K, partial sums meet in LDS, and the epilogue fuses bias/residual (K-free adds), SiLU*mul (K9),
//     interleaved RoPE + bf16 cast + paged-cache scatter (K7 + K1).
#include "common.h"
#include "scratch.h"
#include "../../include/mi355_vllm.h"
#include <string.h>
#include <map>
#include <mutex>
#include <utility>

#define Q4K_BLOCK 144
#define Q6K_BLOCK 210
#define Q4K_TILE (16 * Q4K_BLOCK)   // 2304
#define Q6K_TILE (16 * Q6K_BLOCK)   // 3360
#define BF16_128 0x43004300u        // two bf16 128.0 ; OR-ing a code q < 128 into the mantissa gives 128+q

static inline int tile_bytes_of(int type) { return type == MI355_GGML_Q4_K ? Q4K_TILE : Q6K_TILE; }
static inline int block_bytes_of(int type) { return type == MI355_GGML_Q4_K ? Q4K_BLOCK : Q6K_BLOCK; }

// ================================================================================================
// Host-side repack: native GGUF rows [N][K/256][block] -> tiles [ceil(N/16)][K/256][tile].
//   Q4_K tile (2304 B): hdr[16 rows][16 B = d,dmin,scales[12]] | qs_p0[64 lanes][16 B] | qs_p1[64][16 B]
//       lane = kg*16 + r ; 16 B = { qs[32*(2p)+8kg .. +8] , qs[32*(2p+1)+8kg .. +8] } of row r
//   Q6_K tile (3360 B): sc[16 rows][16 x i8] | ql_n0[64][16 B] | ql_n1[64][16 B] | qh[64][16 B] | d[16 x f16]
//       ql_n lane = { ql[64n+4kg..+4], ql[64n+32+4kg..+4], ql[64n+16+4kg..+4], ql[64n+48+4kg..+4] }
//       qh   lane = { qh[4kg..+4], qh[16+4kg..+4], qh[32+4kg..+4], qh[48+4kg..+4] }
extern "C" int64_t mi355_qweight_repacked_size(int32_t ggml_type, int64_t n_rows, int64_t k) {
    if ((ggml_type != MI355_GGML_Q4_K && ggml_type != MI355_GGML_Q6_K) || k <= 0 || (k % 256) || n_rows <= 0) return -1;
    return ((n_rows + 15) / 16) * (k / 256) * (int64_t)tile_bytes_of(ggml_type);
}

extern "C" int mi355_qweight_repack(void* dst_v, const void* src_v, int32_t ggml_type, int64_t n_rows, int64_t k) {
    if (mi355_qweight_repacked_size(ggml_type, n_rows, k) < 0) return (int)hipErrorInvalidValue;
    const int64_t nkb = k / 256, ntile = (n_rows + 15) / 16;
    uint8_t* dst = static_cast<uint8_t*>(dst_v);
    const uint8_t* src = static_cast<const uint8_t*>(src_v);
    const int bb = block_bytes_of(ggml_type), tb = tile_bytes_of(ggml_type);
#pragma omp parallel for schedule(static)
    for (int64_t rt = 0; rt < ntile; ++rt) {
        for (int64_t kb = 0; kb < nkb; ++kb) {
            uint8_t* t = dst + (rt * nkb + kb) * tb;
            memset(t, 0, tb);
            for (int r = 0; r < 16; ++r) {
                const int64_t row = rt * 16 + r;
                if (row >= n_rows) continue;                       // padded rows stay all-zero (d = 0)
                const uint8_t* b = src + (row * nkb + kb) * bb;
                if (ggml_type == MI355_GGML_Q4_K) {
                    memcpy(t + r * 16, b, 16);
                    const uint8_t* qs = b + 16;
                    for (int p = 0; p < 2; ++p)
                        for (int kg = 0; kg < 4; ++kg) {
                            uint8_t* o = t + 256 + p * 1024 + (kg * 16 + r) * 16;
                            memcpy(o, qs + 32 * (2 * p) + 8 * kg, 8);
                            memcpy(o + 8, qs + 32 * (2 * p + 1) + 8 * kg, 8);
                        }
                } else {
                    const uint8_t *ql = b, *qh = b + 128, *sc = b + 192, *d = b + 208;
                    memcpy(t + r * 16, sc, 16);
                    for (int n = 0; n < 2; ++n)
                        for (int kg = 0; kg < 4; ++kg) {
                            uint8_t* o = t + 256 + n * 1024 + (kg * 16 + r) * 16;
                            memcpy(o + 0, ql + 64 * n + 4 * kg, 4);
                            memcpy(o + 4, ql + 64 * n + 32 + 4 * kg, 4);
                            memcpy(o + 8, ql + 64 * n + 16 + 4 * kg, 4);
                            memcpy(o + 12, ql + 64 * n + 48 + 4 * kg, 4);
                        }
                    for (int kg = 0; kg < 4; ++kg) {
                        uint8_t* o = t + 2304 + (kg * 16 + r) * 16;
                        memcpy(o + 0, qh + 4 * kg, 4);
                        memcpy(o + 4, qh + 16 + 4 * kg, 4);
                        memcpy(o + 8, qh + 32 + 4 * kg, 4);
                        memcpy(o + 12, qh + 48 + 4 * kg, 4);
                    }
                    memcpy(t + 3328 + 2 * r, d, 2);
                }
            }
        }
    }
    return 0;
}

// ================================================================================================
// Reference kernels on the NATIVE GGUF layout (simple, obviously-correct; used for dequantisation at
// load time -- token_embd, quantized_llama.rs:262-264 -- and as an on-device cross-check of the MFMA path).
__device__ __forceinline__ float dequant_native(int type, const uint8_t* __restrict__ blk, int i) {
    if (type == MI355_GGML_Q4_K) {
        const float d = f16_bits_to_f32(*reinterpret_cast<const uint16_t*>(blk));
        const float dmin = f16_bits_to_f32(*reinterpret_cast<const uint16_t*>(blk + 2));
        const uint8_t* s = blk + 4;
        const int j = i >> 5, l = i & 31, g = j >> 1;
        int sc, m;
        if (j < 4) { sc = s[j] & 63; m = s[j + 4] & 63; }
        else { sc = (s[j + 4] & 0xF) | ((s[j - 4] >> 6) << 4); m = (s[j + 4] >> 4) | ((s[j] >> 6) << 4); }
        const uint8_t qb = blk[16 + 32 * g + l];
        const int q = (j & 1) ? (qb >> 4) : (qb & 0xF);
        return d * (float)sc * (float)q - dmin * (float)m;
    } else {
        const uint8_t *ql = blk, *qh = blk + 128;
        const int8_t* sc = reinterpret_cast<const int8_t*>(blk + 192);
        const float d = f16_bits_to_f32(*reinterpret_cast<const uint16_t*>(blk + 208));
        const int n = i >> 7, t = (i >> 5) & 3, l = i & 31;
        const uint8_t lb = ql[64 * n + 32 * (t & 1) + l];
        const int lo4 = (t & 2) ? (lb >> 4) : (lb & 0xF);
        const int hi2 = (qh[32 * n + l] >> (2 * t)) & 3;
        const int q = (lo4 | (hi2 << 4)) - 32;
        return d * (float)sc[8 * n + 2 * t + (l >> 4)] * (float)q;
    }
}

__global__ void __launch_bounds__(256) dequantize_native_kernel(float* __restrict__ out, const uint8_t* __restrict__ w,
                                                                int type, int64_t n_blocks) {
    const int bb = (type == MI355_GGML_Q4_K) ? Q4K_BLOCK : Q6K_BLOCK;
    for (int64_t b = blockIdx.x; b < n_blocks; b += gridDim.x)
        out[b * 256 + threadIdx.x] = dequant_native(type, w + b * bb, threadIdx.x);
}
extern "C" int mi355_dequantize(float* out, const void* w_native, int32_t ggml_type, int64_t n_elems, int64_t stream) {
    if ((ggml_type != MI355_GGML_Q4_K && ggml_type != MI355_GGML_Q6_K) || (n_elems % 256)) return (int)hipErrorInvalidValue;
    if (n_elems == 0) return 0;
    const int64_t nb = n_elems / 256;
    hipLaunchKernelGGL(dequantize_native_kernel, dim3((unsigned)(nb < 65535 ? nb : 65535)), dim3(256), 0,
                       to_stream(stream), out, (const uint8_t*)w_native, ggml_type, nb);
    return (int)hipGetLastError();
}

// one wave per (row, token): y[t][row] = sum_k x[t][k] * dequant(W[row][k])
__global__ void __launch_bounds__(256) qmatmul_ref_kernel(float* __restrict__ out, const float* __restrict__ x,
                                                          const uint8_t* __restrict__ w, int type, int T, int N, int K) {
    const int bb = (type == MI355_GGML_Q4_K) ? Q4K_BLOCK : Q6K_BLOCK;
    const int lane = threadIdx.x & 63;
    const int64_t wid = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wid >= (int64_t)T * N) return;
    const int t = (int)(wid / N), row = (int)(wid % N);
    const uint8_t* wr = w + (int64_t)row * (K / 256) * bb;
    const float* xr = x + (int64_t)t * K;
    float acc = 0.f;
    for (int k = lane; k < K; k += 64) acc = fmaf(xr[k], dequant_native(type, wr + (int64_t)(k >> 8) * bb, k & 255), acc);
    acc = wave_sum(acc);
    if (lane == 0) out[(int64_t)t * N + row] = acc;
}
extern "C" int mi355_qmatmul_ref(float* out, const float* x, const void* w_native, int32_t ggml_type, int32_t T,
                                 int32_t N, int32_t K, int64_t stream) {
    if ((ggml_type != MI355_GGML_Q4_K && ggml_type != MI355_GGML_Q6_K) || (K % 256) || T < 0 || N <= 0) return (int)hipErrorInvalidValue;
    if (T == 0) return 0;
    const int64_t waves = (int64_t)T * N;
    hipLaunchKernelGGL(qmatmul_ref_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, to_stream(stream), out, x,
                       (const uint8_t*)w_native, ggml_type, T, N, K);
    return (int)hipGetLastError();
}

// ================================================================================================
// The MFMA streaming kernel.
struct QmmSeg {
    const uint8_t* w;     // repacked tiles [n_tiles][nkb][tile]
    int32_t type;         // MI355_GGML_Q4_K / Q6_K
    int32_t n_tiles;      // 16-row tiles in this segment
    int32_t row0;         // first output row of this segment (in the concatenated output space)
    int32_t n_rows;       // true rows (<= 16*n_tiles)
};

struct QmmArgs {
    QmmSeg seg[3];
    int32_t nseg;
    int32_t paired;           // 1: seg[0] = gate, seg[1] = up; slot r of a workgroup = tile (r/2) of seg (r&1)
    const void* x;            // [B][ldx] f32 (or bf16 when x_dtype == MI355_DTYPE_BF16)
    int32_t x_dtype;
    int32_t ldx, K, B, kch;   // kch = k-blocks staged per LDS chunk
    const float* norm_w;      // fused RMSNorm weight [K] or null
    float eps;
    int32_t epi;              // MI355_EPI_*
    float* out;               // [B][ldo]
    int32_t ldo;
    const float* resid;       // [B][ldo] (EPI_RESID), may alias out
    const float* bias;        // [sum rows] or null
    // EPI_QKV_ROPE_CACHE
    const float* cos_t;
    const float* sin_t;
    const int64_t* positions;
    const int64_t* slot_mapping;
    uint16_t* q_out;          // bf16 [B][Hq*D]
    uint16_t* kcache;
    uint16_t* vcache;
    int32_t Hq, Hkv, D, rot, block_size, kv_layout;
    int32_t dbg;              // experiments: 1 = stream the weights but skip unpack/MFMA (memory-path probe)
    // chain request (wide path): the epilogue also stages the activation image of the NEXT mat-mul, whose x is this
    // `out` with k = next_k and RMSNorm weight next_norm_w (or null)
    const float* next_norm_w;
    int32_t chain_next, next_k;
    // mixture of experts (quantized_llama.rs:56-123 on the device): blockIdx.y = (token, slot) pair; the pair's expert
    // id is read from device memory and selects the weight slab, its x row is pair / moe_xdiv, its out row is pair
    const int32_t* moe_expert;
    int32_t moe_pairs, moe_xdiv;
    const int32_t* rows_dev; int32_t rows_min;   // wide path: no-op unless *rows_dev > rows_min (mi355_qmm_desc)
    int64_t moe_stride[3];    // bytes between consecutive experts of each segment
    // grouped launch (9..32-token path): grp_n independent mat-muls of these shapes; group e reads x + e * grp_x, weights + e *
    // moe_stride[s], writes out + e * grp_out (elements), gated by rows_dev[e]
    int32_t grp_n;
    int64_t grp_x, grp_out;
    // prompt-step grouped launch over a device block table (mi355_qmm_desc.group_block_table): token block b = {expert, row end}
    const int32_t* blk_tab;
};

struct TileRegs { uint4 a, b, c, d; uint32_t e; };
// Probe / ablation modes (mi355_set_tuning(2, v)) are compiled into -DMI355_QMM_PROBES builds only (tools/ build their own
// library): a guarded load or branch in a hot loop costs the production kernels measurable time.
#ifdef MI355_QMM_PROBES
#define QMM_DBG(a) ((a).dbg)
#else
#define QMM_DBG(a) 0
#endif
typedef const void __attribute__((address_space(1)))* qmg_gptr_t;
typedef void __attribute__((address_space(3)))* qmg_lptr_t;
// one wave-wide 16-B LDS DMA (lane i lands at lds_dst + 16 i) the compiler does NOT track: the caller counts s_waitcnt vmcnt itself
__device__ __forceinline__ void qmg_dma16(const uint8_t* gsrc_lane, uint32_t lds_dst) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc_lane), "s"(lds_dst) : "memory");
}

// WT = MI355_GGML_Q4_K / MI355_GGML_Q6_K when every tile of the launch has that type (3 resp. 5 loads per unit),
// 0 for mixed launches: both types then issue 5 loads (Q4_K adds two same-line dummies) so that the number of
// outstanding loads is a compile-time constant and the compiler can emit COUNTED s_waitcnt vmcnt(N).
// Weights are streamed once -> non-temporal loads.
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ld_nt16(const uint4* p) {
    const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}

template <int WT>
__device__ __forceinline__ TileRegs load_tile(int type, const uint8_t* __restrict__ t, int lane) {
    TileRegs r;
    const int row = lane & 15;
    const uint4* t4 = reinterpret_cast<const uint4*>(t);
    r.a = ld_nt16(t4 + row);                    // Q4_K: d,dmin,scales[12] ; Q6_K: 16 x i8 scales
    r.b = ld_nt16(t4 + 16 + lane);              // Q4_K: groups 0,1 ; Q6_K: ql half 0
    r.c = ld_nt16(t4 + 80 + lane);              // Q4_K: groups 2,3 ; Q6_K: ql half 1
    if (WT == MI355_GGML_Q4_K) {
        r.d = make_uint4(0, 0, 0, 0);
        r.e = 0;
    } else if (WT == MI355_GGML_Q6_K) {
        r.d = ld_nt16(t4 + 144 + lane);         // qh
        r.e = __builtin_nontemporal_load(reinterpret_cast<const uint16_t*>(t + 3328) + row);   // d (f16)
    } else {
        const bool q6 = (type == MI355_GGML_Q6_K);
        r.d = ld_nt16(t4 + (q6 ? 144 + lane : 0));
        r.e = __builtin_nontemporal_load(reinterpret_cast<const uint16_t*>(t + (q6 ? 3328 : 0)) + (q6 ? row : 0));
    }
    return r;
}

// LDS image of the staged activations of one k-block (BT batch entries), private to a wave:
//   ximg : [32 entries][2*BT rows][8 bf16]  entry E <-> elements 8E..8E+7 in order {0,2,1,3,4,6,5,7};
//          rows 0..BT-1 = hi(bf16(x)), rows BT..2BT-1 = lo(bf16(x - hi))
// A-fragment row of a lane: MFMA rows 0..7 carry hi(x) of batch 0..7, rows 8..15 carry lo(x).  Rows beyond
// the staged batch (m >= BT) just re-read a staged row: MFMA rows are independent and those outputs are
// never consumed, so no zero-fill and no branch is needed.
template <int BT>
__device__ __forceinline__ int a_row_of(int m) {
    const int mm = (m & 7) < BT ? (m & 7) : (BT - 1);
    return (m < 8) ? mm : BT + mm;
}

// ---- activations: every wave stages the k-block it is about to consume into its OWN 1.2*BT KB of LDS
// (no workgroup barrier, no prologue: the first x loads leave together with the first weight loads).
// Lane L of the wave owns elements 4L..4L+3 of the k-block for every batch row.
template <int BT>
struct XRegs {
    uint4 v[BT];      // f32 x: 4 floats ; bf16 x: .x/.y hold 4 bf16
    float4 nw;        // RMSNorm weight of the 4 elements (1 when no norm is fused)
};

// XB: 1 = x is bf16, 0 = f32 (the single-launch kernels are built for both: a run-time branch around the activation loads
// of the hot loop measured 2.3 % of the batch-1 step), -1 = decide at run time
// NRM: 1 = an RMSNorm weight is fused, 0 = none (no weight load, no sum of squares), -1 = decide at run time -- the same reason
// (the branch in the staging step measured 1.4 % of the batch-1 step)
// what a (token, slot) pair of a mixture-of-experts launch adds to the descriptor's pointers: wave-uniform values next to the
// kernarg descriptor, applied under `if constexpr (MOE)` only: with the offsets always present (zeros in the dense kernels) the
// dense Q4_K f32-x kernel came out of hipcc producing NaN at several wave counts (tools/dbg_nan.py) although its ISA differed
// from the good one by a handful of scalar instructions -- the dense kernels are kept byte-identical to what was validated.  The first form of qmm_moe_kernel patched a private
// COPY of the descriptor instead: 312 bytes of scratch per lane, every argument a scratch load -- the expert mat-vecs ran at
// 2.0-2.3 TB/s where the dense ones reach 3.5-3.9.
struct QmmPairOff { size_t w0, w1, w2; size_t x_bytes; size_t out; };
template <int BT, int XB = -1, int NRM = -1>
__device__ __forceinline__ XRegs<BT> load_x(const QmmArgs& a, int kb, int lane, const float* nwp, const size_t x_bytes = 0) {
    const uint8_t* xb0 = static_cast<const uint8_t*>(a.x) + x_bytes;
    XRegs<BT> r;
    const size_t k = (size_t)kb * 256 + 4 * lane;
#pragma unroll
    for (int b = 0; b < BT; ++b) {
        const int bb = (BT == 1 || b < a.B) ? b : a.B - 1;        // padded rows re-read the last row (zeroed later)
        if (XB < 0 ? a.x_dtype == MI355_DTYPE_BF16 : XB == 1) {
            const uint2 t = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(xb0) + (size_t)bb * a.ldx + k);
            r.v[b] = make_uint4(t.x, t.y, 0, 0);
        } else {
            r.v[b] = *reinterpret_cast<const uint4*>(reinterpret_cast<const float*>(xb0) + (size_t)bb * a.ldx + k);
        }
    }
    if (NRM != 0) r.nw = *reinterpret_cast<const float4*>(nwp + k);
    else r.nw = make_float4(1.f, 1.f, 1.f, 1.f);
    return r;
}

// ================= inner math of the decode mat-vec ============================================================
// The decode mat-vec is VALU-issue bound, not HBM bound (rocprofv3: VALU pipe ~80 % busy over a wave's lifetime,
// same duration with the weights resident in the Infinity Cache), so the inner loop is trimmed to the essentials:
//   * the "+128" of the bf16 code trick is cancelled INSIDE the MFMA: one extra MFMA per sub-block with a constant
//     B operand of -128 (Q6_K: -160) yields C_in = -128*sum(x) in accumulator layout; the weight MFMA then starts
//     from C_in and returns the true dot product.  The same C_in is the sub-block sum the Q4_K minimum term needs
//     -> no shuffle reductions and no sum arrays in the staging step, no per-sub-block offset FMAs;
//   * d and dmin are applied once per tile (sum_j sc_j*P_j and sum_j m_j*S_j are formed first);
//   * tiles of one k-block that share a type are computed together so C_in and the A fragments are shared.
template <int BT, int XB = -1, int NRM = -1>
__device__ __forceinline__ void stage_kblock3(const QmmArgs& a, const XRegs<BT>& xr, uint8_t* ximg, int lane, float (&ss)[BT]) {
    const int E = lane >> 1, half = lane & 1;
#pragma unroll
    for (int b = 0; b < BT; ++b) {
        float v[4];
        if (XB < 0 ? a.x_dtype == MI355_DTYPE_BF16 : XB == 1) {
            v[0] = bf16lo_to_f32(xr.v[b].x); v[1] = bf16hi_to_f32(xr.v[b].x);
            v[2] = bf16lo_to_f32(xr.v[b].y); v[3] = bf16hi_to_f32(xr.v[b].y);
        } else {
            v[0] = __uint_as_float(xr.v[b].x); v[1] = __uint_as_float(xr.v[b].y);
            v[2] = __uint_as_float(xr.v[b].z); v[3] = __uint_as_float(xr.v[b].w);
        }
        if (BT > 1 && b >= a.B) { v[0] = v[1] = v[2] = v[3] = 0.f; }
        if (NRM < 0 ? a.norm_w != nullptr : NRM == 1) {
            ss[b] = fmaf(v[0], v[0], fmaf(v[1], v[1], fmaf(v[2], v[2], fmaf(v[3], v[3], ss[b]))));
            v[0] *= xr.nw.x; v[1] *= xr.nw.y; v[2] *= xr.nw.z; v[3] *= xr.nw.w;
        }
        const uint32_t h02 = cvt_pk_bf16(v[0], v[2]), h13 = cvt_pk_bf16(v[1], v[3]);
        const uint32_t l02 = cvt_pk_bf16(v[0] - bf16lo_to_f32(h02), v[2] - bf16hi_to_f32(h02));
        const uint32_t l13 = cvt_pk_bf16(v[1] - bf16lo_to_f32(h13), v[3] - bf16hi_to_f32(h13));
        *reinterpret_cast<uint2*>(ximg + ((size_t)E * (2 * BT) + b) * 16 + half * 8) = make_uint2(h02, h13);
        *reinterpret_cast<uint2*>(ximg + ((size_t)E * (2 * BT) + BT + b) * 16 + half * 8) = make_uint2(l02, l13);
    }
}

template <int BT, int NV, int RR>
__device__ __forceinline__ void compute3_q4k(const TileRegs* w, const uint8_t* ximg, int lane, float (*y)[NV]) {
    const int m = lane & 15, kg = lane >> 4;
    const uint8_t* abase = ximg + ((size_t)kg * (2 * BT) + a_row_of<BT>(m)) * 16;
    uint4 aw[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) aw[j] = *reinterpret_cast<const uint4*>(abase + (size_t)j * 4 * (2 * BT) * 16);
    float d[RR], dm[RR];
    uint32_t scl[RR], sch[RR], mnl[RR], mnh[RR];
#pragma unroll
    for (int r = 0; r < RR; ++r) {
        d[r] = f16_bits_to_f32((uint16_t)(w[r].a.x & 0xFFFF));
        dm[r] = f16_bits_to_f32((uint16_t)(w[r].a.x >> 16)) * (1.f / 128.f);
        const uint32_t s0 = w[r].a.y, s1 = w[r].a.z, s2 = w[r].a.w;
        scl[r] = s0 & 0x3F3F3F3Fu;
        mnl[r] = s1 & 0x3F3F3F3Fu;
        sch[r] = (s2 & 0x0F0F0F0Fu) | ((s0 >> 2) & 0x30303030u);
        mnh[r] = ((s2 >> 4) & 0x0F0F0F0Fu) | ((s1 >> 2) & 0x30303030u);
        // opaque: four 6-bit values per word, one v_cvt_f32_ubyteN each (no per-byte re-masking)
        asm volatile("" : "+v"(scl[r]), "+v"(sch[r]), "+v"(mnl[r]), "+v"(mnh[r]));
    }
    uint32_t nib = 0x000F000Fu, n128 = 0xC300C300u;              // bf16 -128.0 twice
    asm volatile("" : "+v"(nib), "+v"(n128));
    const uint4 negw = make_uint4(n128, n128, n128, n128);
    float sP[RR][NV], sM[RR][NV];
#pragma unroll
    for (int r = 0; r < RR; ++r)
#pragma unroll
        for (int v = 0; v < NV; ++v) { sP[r][v] = 0.f; sM[r][v] = 0.f; }
    const f32x4_t zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const bf16x8_t af = __builtin_bit_cast(bf16x8_t, aw[j]);
        const f32x4_t cin = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, __builtin_bit_cast(bf16x8_t, negw), zero, 0, 0, 0);
        const int p = j >> 2, pr = (j >> 1) & 1, sh = (j & 1) * 4;
#pragma unroll
        for (int r = 0; r < RR; ++r) {
            const uint4 qs = p ? w[r].c : w[r].b;
            const uint32_t w0 = pr ? qs.z : qs.x, w1 = pr ? qs.w : qs.y;
            uint4 bw;
            bw.x = ((w0 >> sh) & nib) | BF16_128;                // elements (b0, b2)
            bw.y = ((w0 >> (sh + 8)) & nib) | BF16_128;          // elements (b1, b3)
            bw.z = ((w1 >> sh) & nib) | BF16_128;                // elements (b4, b6)
            bw.w = ((w1 >> (sh + 8)) & nib) | BF16_128;          // elements (b5, b7)
            const f32x4_t acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, __builtin_bit_cast(bf16x8_t, bw), cin, 0, 0, 0);
            const float sc = (float)((((j < 4) ? scl[r] : sch[r]) >> (8 * (j & 3))) & 0xFF);
            const float mn = (float)((((j < 4) ? mnl[r] : mnh[r]) >> (8 * (j & 3))) & 0xFF);
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                sP[r][v] = fmaf(sc, acc[v], sP[r][v]);
                sM[r][v] = fmaf(mn, cin[v], sM[r][v]);           // cin = -128 * sum(x): minimum term, sign folded
            }
        }
    }
#pragma unroll
    for (int r = 0; r < RR; ++r)
#pragma unroll
        for (int v = 0; v < NV; ++v) y[r][v] = fmaf(d[r], sP[r][v], fmaf(dm[r], sM[r][v], y[r][v]));
}

template <int BT, int NV, int RR>
__device__ __forceinline__ void compute3_q6k(const TileRegs* w, const uint8_t* ximg, int lane, float (*y)[NV]) {
    const int m = lane & 15, kg = lane >> 4;
    const uint8_t* abase = ximg + ((size_t)(kg >> 1) * (2 * BT) + a_row_of<BT>(m)) * 16 + (kg & 1) * 8;
    uint32_t bmask = 0x00FF00FFu, n160 = 0xC320C320u;            // bf16 -160.0 twice (code = q + 32, +128)
    asm volatile("" : "+v"(bmask), "+v"(n160));
    const uint2 negw = make_uint2(n160, n160);
    float sP[RR][NV];
#pragma unroll
    for (int r = 0; r < RR; ++r)
#pragma unroll
        for (int v = 0; v < NV; ++v) sP[r][v] = 0.f;
    const f32x4_t zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        uint2 aw[8];                                                  // A fragments of the 8 sub-blocks of this half
#pragma unroll
        for (int s = 0; s < 8; ++s) aw[s] = *reinterpret_cast<const uint2*>(abase + (size_t)(8 * n + s) * 2 * (2 * BT) * 16);
#pragma unroll
        for (int is = 0; is < 2; ++is) {
            uint32_t t[RR][4];
#pragma unroll
            for (int r = 0; r < RR; ++r) {
                const uint4 ql = n ? w[r].c : w[r].b;
                const uint32_t qa = is ? ql.z : ql.x, qb = is ? ql.w : ql.y;
                const uint32_t h = (n ? (is ? w[r].d.w : w[r].d.z) : (is ? w[r].d.y : w[r].d.x));
                t[r][0] = (qa & 0x0F0F0F0Fu) | ((h << 4) & 0x30303030u);
                t[r][1] = (qb & 0x0F0F0F0Fu) | ((h << 2) & 0x30303030u);
                t[r][2] = ((qa >> 4) & 0x0F0F0F0Fu) | (h & 0x30303030u);
                t[r][3] = ((qb >> 4) & 0x0F0F0F0Fu) | ((h >> 2) & 0x30303030u);
            }
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                const int s = 8 * n + 2 * tt + is;                    // 16-sub-block index 0..15
                const s16x4_t af = __builtin_bit_cast(s16x4_t, aw[2 * tt + is]);
                const f32x4_t cin = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(af, __builtin_bit_cast(s16x4_t, negw), zero, 0, 0, 0);
#pragma unroll
                for (int r = 0; r < RR; ++r) {
                    uint2 bw;
                    bw.x = (t[r][tt] & bmask) | BF16_128;                 // elements (b0, b2)
                    bw.y = ((t[r][tt] >> 8) & bmask) | BF16_128;          // elements (b1, b3)
                    const f32x4_t acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(af, __builtin_bit_cast(s16x4_t, bw), cin, 0, 0, 0);
                    const uint32_t scw = (s >> 2) == 0 ? w[r].a.x : ((s >> 2) == 1 ? w[r].a.y : ((s >> 2) == 2 ? w[r].a.z : w[r].a.w));
                    const float sc = (float)(int)(int8_t)((scw >> (8 * (s & 3))) & 0xFF);
#pragma unroll
                    for (int v = 0; v < NV; ++v) sP[r][v] = fmaf(sc, acc[v], sP[r][v]);
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < RR; ++r) {
        const float d = f16_bits_to_f32((uint16_t)(w[r].e & 0xFFFF));
#pragma unroll
        for (int v = 0; v < NV; ++v) y[r][v] = fmaf(d, sP[r][v], y[r][v]);
    }
}

__device__ __forceinline__ float silu_f(float g) { return g / (1.f + __expf(-g)); }

#ifndef QMM_PF_MIN
#define QMM_PF_MIN 2   // weight-prefetch ring depth in (k-block, tile) units per wave = max(R, QMM_PF_MIN)
#endif

// One workgroup = R row tiles x all of K.  Its NW waves split the k-blocks (wave w owns k-blocks w, w+NW, ...);
// every wave walks its (k-block, tile) units through a QMM_PF-deep register ring.  The ring loop is fully
// static (unrolled over the ring slots, loads issued unconditionally -- past the end they hit one dummy line)
// so the compiler emits counted vmcnt waits and ~QMM_PF KiB-sized loads per lane stay in flight.  There is no
// workgroup-level prologue: the only barrier is the one in front of the epilogue.
// per-lane segment index -> fields by select over the (<= 3) segments: the descriptor stays in SGPRs (a runtime-indexed
// a.seg[i] would be a vector load from the kernarg segment on the critical path)
__device__ __forceinline__ int seg_pick(int v0, int v1, int v2, int i) {
    v0 = __builtin_amdgcn_readfirstlane(v0); v1 = __builtin_amdgcn_readfirstlane(v1); v2 = __builtin_amdgcn_readfirstlane(v2);
    asm volatile("" : "+s"(v0), "+s"(v1), "+s"(v2));              // opaque SGPRs: the selects must not be re-fused into a load
    return i == 0 ? v0 : (i == 1 ? v1 : v2);
}
__device__ __forceinline__ int seg_n_rows(const QmmArgs& a, int i) { return seg_pick(a.seg[0].n_rows, a.seg[1].n_rows, a.seg[2].n_rows, i); }
__device__ __forceinline__ int seg_row0(const QmmArgs& a, int i) { return seg_pick(a.seg[0].row0, a.seg[1].row0, a.seg[2].row0, i); }

// ---- epilogue operands that do not depend on the mat-vec (residual, RoPE cos/sin, cache slot) are fetched under the
// weight stream by the threads that will write the outputs: `early` goes out ahead of the first weight loads (the
// position, the slot, the residual), `late` right after them (cos/sin need the position)
struct EpiPre { int64_t pos, slot; float f0, f1; int lrow, sgi; bool live, rope; int64_t spos, sslot; };
template <int BT, int R>
__device__ __forceinline__ void epi_pre_early(const QmmArgs& a, const int (&segi)[R], const int (&tile)[R], EpiPre& ep) {
    constexpr int NOUT = R * BT * 16;
    const int e_rr = threadIdx.x & 15, e_b = ((int)threadIdx.x >> 4) % BT, e_r = (int)threadIdx.x / (16 * BT);
    int e_tl = 0;
    ep.sgi = 0;
#pragma unroll
    for (int q = 0; q < R; ++q) if (q == e_r) { ep.sgi = segi[q]; e_tl = tile[q]; }
    ep.lrow = e_tl * 16 + e_rr;
    ep.live = (int)threadIdx.x < NOUT && e_b < a.B && ep.lrow < seg_n_rows(a, ep.sgi);
    ep.rope = ep.live && a.epi == MI355_EPI_QKV_ROPE_CACHE;
    ep.pos = 0; ep.slot = -1; ep.f0 = 0.f; ep.f1 = 0.f; ep.spos = 0; ep.sslot = -1;   // RESID: f0 = residual value ; ROPE: f0, f1 = cos, sin
#ifndef QMM_SCALAR_POS
#define QMM_SCALAR_POS 0      // measured: scalar position / slot loads made the batch-1 step 1.4 % SLOWER (1.960 vs 1.933 ms, one box, alternated): the
#endif                         // extra uniform branch and SGPR pairs sit in every variant of the kernel, the stall they remove in one wave of q|k|v
    if (QMM_SCALAR_POS && BT == 1) {
        // one token: position and slot are wave-uniform -- SCALAR loads that leave with the kernarg burst's registers instead of vector
        // loads whose round trip the cos / sin request (epi_pre_late) then waits for in front of the wave's first staging step
        // (profiles/r06_b1_inside_launch.txt: the wave that holds the epilogue lanes was the straggler of every q|k|v workgroup)
        if (a.epi == MI355_EPI_QKV_ROPE_CACHE) {
            // (through the constant address space: hipcc keeps a uniform load from plain global memory on the vector unit and waits
            // for it at once; both arrays were written by an EARLIER launch, so the scalar cache's view is the current one)
            typedef const int64_t __attribute__((address_space(4))) * cptr_t;
            ep.spos = *reinterpret_cast<cptr_t>(reinterpret_cast<uintptr_t>(a.positions));
            ep.sslot = *reinterpret_cast<cptr_t>(reinterpret_cast<uintptr_t>(a.slot_mapping));
        }
    } else {
        if (ep.rope && ep.sgi < 2) ep.pos = a.positions[e_b];
        if (ep.rope && ep.sgi != 0) ep.slot = a.slot_mapping[e_b];
    }
    if (ep.live && a.epi == MI355_EPI_RESID) ep.f0 = a.resid[(size_t)e_b * a.ldo + seg_row0(a, ep.sgi) + ep.lrow];
    __builtin_amdgcn_sched_barrier(0);                             // keep these loads ahead of the weight loads (in-order vmcnt)
}
template <int BT>
__device__ __forceinline__ void epi_pre_late(const QmmArgs& a, EpiPre& ep) {
    if (QMM_SCALAR_POS && BT == 1) {
        if (a.epi == MI355_EPI_QKV_ROPE_CACHE) {
            if (ep.rope && ep.sgi < 2) ep.pos = ep.spos;
            if (ep.rope && ep.sgi != 0) ep.slot = ep.sslot;
        }
    }
    if (ep.rope && ep.sgi < 2) {
        const int d = ep.lrow % a.D;
        if (d < a.rot) {
            ep.f0 = a.cos_t[ep.pos * (a.rot >> 1) + (d >> 1)];
            ep.f1 = a.sin_t[ep.pos * (a.rot >> 1) + (d >> 1)];
        }
    }
}

// red: [NW][R][BT][16] partial sums of the NW compute waves, red_ss: [NW][BT] partial sum x^2 (fused RMSNorm)
template <int BT, int R>
__device__ __forceinline__ void qmm_epilogue(const QmmArgs& a, const float* red, const float* red_ss, const int NW,
                                             const int (&segi)[R], const int (&tile)[R], const EpiPre& ep, const size_t out_off = 0) {
    float* const aout = a.out + out_off;
    // ---- epilogue: thread -> (slot r, batch b, row rr)
    constexpr int NOUT = R * BT * 16;
    const int nout = NOUT;
    for (int idx = threadIdx.x; idx < nout; idx += blockDim.x) {
        const int rr = idx & 15, b = (idx >> 4) % BT, r = idx / (16 * BT);
#ifdef QMM_NO_EPI_PRE
        const bool pre = false;
#else
        const bool pre = idx == (int)threadIdx.x;                   // first pass: operands were prefetched
#endif
        if (b >= a.B) continue;
        float rs = 1.f;
        if (a.norm_w) {
            float t = 0.f;
            for (int w = 0; w < NW; ++w) t += red_ss[w * BT + b];
            rs = rsqrtf(t / (float)a.K + a.eps);
        }
        auto sum_of = [&](int slot, int rowi) {
            float s = 0.f;
            for (int w = 0; w < NW; ++w) s += red[(((size_t)w * R + slot) * BT + b) * 16 + rowi];
            return s * rs;
        };
        // runtime slot index (r) only touches scalars here
        int sgi = 0, tl = 0;
#pragma unroll
        for (int q = 0; q < R; ++q) if (q == r) { sgi = segi[q]; tl = tile[q]; }
        const int lrow = tl * 16 + rr;                             // row inside the segment
        if (lrow >= seg_n_rows(a, sgi)) continue;
        const int orow = seg_row0(a, sgi) + lrow;
        if (a.epi == MI355_EPI_SILU_MUL) {
            if (r & 1) continue;                                   // even slot = gate, odd slot = up of the same rows
            float g = sum_of(r, rr), up = sum_of(r + 1, rr);
            if (a.bias) { g += a.bias[orow]; up += a.bias[a.seg[1].row0 + lrow]; }
            aout[(size_t)b * a.ldo + lrow] = silu_f(g) * up;
            continue;
        }
        float val = sum_of(r, rr);
        if (a.bias) val += a.bias[orow];
        if (a.epi == MI355_EPI_STORE) {
            aout[(size_t)b * a.ldo + orow] = val;
        } else if (a.epi == MI355_EPI_RESID) {
            aout[(size_t)b * a.ldo + orow] = (pre ? ep.f0 : a.resid[(size_t)b * a.ldo + orow]) + val;
        } else if (a.epi == MI355_EPI_QKV_ROPE_CACHE) {
            // segment 0 = q, 1 = k, 2 = v ; interleaved RoPE on (even,odd) channel pairs of q and k
            const int D = a.D, d = lrow % D, hh = lrow / D;
            float o = val;
            if (sgi < 2 && d < a.rot) {
                float pv = sum_of(r, rr ^ 1);
                if (a.bias) pv += a.bias[orow ^ 1];
                float c = ep.f0, s = ep.f1;
                if (!pre) {
                    const int64_t pos = a.positions[b];
                    c = a.cos_t[pos * (a.rot >> 1) + (d >> 1)];
                    s = a.sin_t[pos * (a.rot >> 1) + (d >> 1)];
                }
                o = (d & 1) ? (pv * s + val * c) : (val * c - pv * s);
            }
            const uint16_t ob = f32_to_bf16(o);
            if (sgi == 0) {
                a.q_out[(size_t)b * a.Hq * D + lrow] = ob;
            } else {
                const int64_t slot = pre ? ep.slot : a.slot_mapping[b];
                if (slot >= 0) {
                    uint16_t* cache = (sgi == 1) ? a.kcache : a.vcache;
                    if (a.kv_layout == MI355_KV_PAGED_FP8) {        // e4m3fn of the bf16 value, K layout x = 16
                        uint8_t* c8 = reinterpret_cast<uint8_t*>(cache);
                        const int64_t blk = slot / a.block_size, off = slot % a.block_size;
                        const uint8_t q8 = to_e4m3(bf16_to_f32(ob));
                        if (sgi == 1) c8[((((blk * a.Hkv + hh) * (D / 16) + d / 16) * a.block_size + off) * 16) + d % 16] = q8;
                        else c8[((blk * a.Hkv + hh) * D + d) * (int64_t)a.block_size + off] = q8;
                    } else if (a.kv_layout == MI355_KV_FLASH) {
                        cache[(slot * a.Hkv + hh) * D + d] = ob;
                    } else {
                        const int64_t blk = slot / a.block_size, off = slot % a.block_size;
                        if (sgi == 1)
                            cache[((((blk * a.Hkv + hh) * (D / 8) + d / 8) * a.block_size + off) * 8) + d % 8] = ob;
                        else
                            cache[((blk * a.Hkv + hh) * D + d) * (int64_t)a.block_size + off] = ob;
                    }
                }
            }
        }
    }
}

// experiments only (probe mode 7): per-wave timestamps [workgroup][wave 16][4] = entry, main loop done, past the barrier, exit.
// The hooks are compiled in ONLY with -DMI355_QMM_TIMESTAMPS (tools/exp_wave_times.py builds its own library): four dead
// scalar branches in the single-token kernel measured 463 vs 485 tok/s at batch 1.
__device__ unsigned long long* g_qmm_ts = nullptr;
extern "C" int mi355_debug_set_timestamps(void* dev_ptr) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_qmm_ts), &dev_ptr, sizeof(dev_ptr));
}
#ifdef MI355_QMM_TIMESTAMPS
__device__ __forceinline__ void qmm_stamp(const QmmArgs& a, int i) {
    if (a.dbg == 7 && (threadIdx.x & 63) == 0 && g_qmm_ts)
        g_qmm_ts[((size_t)blockIdx.x * 16 + (threadIdx.x >> 6)) * 4 + i] = wall_clock64();
}
#else
#define qmm_stamp(a, i) ((void)0)
#endif
// Round 6: the INSIDE of a single-token launch on a timeline (-DMI355_QMM_TIMELINE builds only, tools/exp_b1_timeline.py):
// eight 100 MHz timestamps per wave, kept in registers and written once at exit (a store in the loop would join the counted vmcnt waits):
//   0 wave entry   1 kernarg fields in SGPRs   2 every load of the prologue issued   3 first k-block staged (= x and norm weight arrived)
//   4 first (k-block, tile) unit computed (= first weights arrived)   5 main loop done   6 past the workgroup barrier   7 epilogue stores issued
#ifdef MI355_QMM_TIMELINE
#define QMM_TL_DECL unsigned long long qtl[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define QMM_TL(i) do { qtl[i] = wall_clock64(); } while (0)
#define QMM_TL_ONCE(i) do { if (qtl[i] == 0) qtl[i] = wall_clock64(); } while (0)
#define QMM_TL_FLUSH() do { if ((threadIdx.x & 63) == 0 && g_qmm_ts) { \
        unsigned long long* d_ = g_qmm_ts + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 16 + (threadIdx.x >> 6)) * 8; \
        for (int i_ = 0; i_ < 8; ++i_) d_[i_] = qtl[i_]; } } while (0)
#else
#define QMM_TL_DECL ((void)0)
#define QMM_TL(i) ((void)0)
#define QMM_TL_ONCE(i) ((void)0)
#define QMM_TL_FLUSH() ((void)0)
#endif

// Every descriptor field the single-token kernel touches, requested in ONE burst of scalar loads at wave entry (round 6).  Left to
// itself hipcc loads a kernarg field in the basic block that first uses it: the prologue of qmm_kernel<1,1,...> was a chain of EIGHT
// `s_load -> s_waitcnt lgkmcnt(0) -> branch` round trips (paired, nseg, the segment walk, the tile pointer, B, n_rows, epi, the positions
// pointer, norm_w / K, the x pointer) before the first weight request left the wave -- at the start of a launch every one of them misses
// the scalar cache (the kernarg lines are cold in this XCD's L2 as well), and 160 launches per step pay it.  One asm statement with all
// fields as SGPR inputs makes every load dominate its uses, so the compiler issues them back to back behind a single wait and re-uses
// the registers in the branches below (kernarg memory is invariant: the later a.field reads are the same values).
#ifndef QMM_KARG_BURST
#define QMM_KARG_BURST 1
#endif
#ifndef QMM_SEG_SELECT
#define QMM_SEG_SELECT QMM_KARG_BURST
#endif
__device__ __forceinline__ void qmm_kernarg_burst(const QmmArgs& a) {
#if QMM_KARG_BURST == 2      // only what stands between wave entry and the first weight / activation request
    asm volatile("" ::"s"(a.seg[0].w), "s"(a.seg[1].w), "s"(a.seg[2].w), "s"(a.seg[0].n_tiles), "s"(a.seg[1].n_tiles), "s"(a.nseg), "s"(a.paired),
                 "s"(a.x), "s"(a.K), "s"(a.B), "s"(a.norm_w), "s"(a.epi), "s"((int)blockDim.x));
#elif QMM_KARG_BURST
    asm volatile("" ::"s"(a.seg[0].w), "s"(a.seg[1].w), "s"(a.seg[2].w), "s"(a.seg[0].n_tiles), "s"(a.seg[1].n_tiles), "s"(a.seg[2].n_tiles),
                 "s"(a.seg[0].type), "s"(a.seg[1].type), "s"(a.seg[2].type), "s"(a.seg[0].n_rows), "s"(a.seg[1].n_rows), "s"(a.seg[2].n_rows),
                 "s"(a.seg[0].row0), "s"(a.seg[1].row0), "s"(a.seg[2].row0), "s"(a.nseg), "s"(a.paired), "s"(a.x), "s"(a.ldx), "s"(a.K),
                 "s"(a.B), "s"(a.norm_w), "s"(a.eps), "s"(a.epi), "s"(a.out), "s"(a.ldo), "s"(a.resid), "s"(a.bias), "s"((int)blockDim.x));
    asm volatile("" ::"s"(a.cos_t), "s"(a.sin_t), "s"(a.positions), "s"(a.slot_mapping), "s"(a.q_out), "s"(a.kcache), "s"(a.vcache), "s"(a.Hq),
                 "s"(a.Hkv), "s"(a.D), "s"(a.rot), "s"(a.block_size), "s"(a.kv_layout));
#endif
}

// the same for the kernels of the 9..32-token path (staging, GEMM, split-K epilogue): their prologues walk rows_dev pointer -> wait ->
// *rows_dev -> wait -> three to five further fields, each behind its own wait; here EVERY descriptor field goes out in one burst
// (the row gate's own load follows as the single dependent round trip it has to be)
#ifndef QMM_KARG_BURST_EPI
#define QMM_KARG_BURST_EPI 0       // A/B: the burst in the split-K epilogue / staging launches only (latency chains of a few microseconds)
#endif
#ifndef QMM_KARG_BURST_WIDE
#define QMM_KARG_BURST_WIDE 0      // measured: the full-descriptor burst needs > 100 SGPRs at once, hipcc splits it into three waits, and the ragged batch-32 step LOST 1.6 % (6230 vs 6330 tok/s, profiles/r06_b32_kernarg_burst_ab.txt)
#endif
template <bool EPI = false>
__device__ __forceinline__ void qmm_kernarg_burst_all(const QmmArgs& a) {
#if QMM_KARG_BURST_WIDE || QMM_KARG_BURST_EPI
    if (!(QMM_KARG_BURST_WIDE || (EPI && QMM_KARG_BURST_EPI))) return;
    qmm_kernarg_burst(a);
    asm volatile("" ::"s"(a.next_norm_w), "s"(a.chain_next), "s"(a.next_k), "s"(a.rows_dev), "s"(a.rows_min), "s"(a.x_dtype), "s"(a.grp_n),
                 "s"(a.grp_x), "s"(a.grp_out), "s"(a.moe_stride[0]), "s"(a.moe_stride[1]), "s"(a.moe_stride[2]));
#endif
}
#if QMM_KARG_BURST_WIDE
#define QMM_BURST_CHAIN(ch) asm volatile("" ::"s"((ch).img), "s"((ch).ssp), "s"((ch).norm_w), "s"((ch).K), "s"((ch).MT), "s"((ch).kbb), "s"((ch).sp))
#define QMM_BURST_VALS(...) qmm_burst_vals(__VA_ARGS__)
template <typename T>
__device__ __forceinline__ int qmm_burst_one(T v) { asm volatile("" ::"s"(v)); return 0; }
template <typename... T>
__device__ __forceinline__ void qmm_burst_vals(T... v) { const int u_[] = {qmm_burst_one(v)...}; (void)u_; }
#else
#define QMM_BURST_CHAIN(ch) ((void)0)
#define QMM_BURST_VALS(...) ((void)0)
#endif

template <int BT, int R, int WT, int XB, int NRM, bool MOE = false>
__device__ __forceinline__ void qmm_body(const QmmArgs& a, const QmmPairOff po = QmmPairOff{0, 0, 0, 0, 0}) {
    qmm_stamp(a, 0);
    QMM_TL_DECL;
    QMM_TL(0);
    qmm_kernarg_burst(a);
    QMM_TL(1);
    // probe modes of this kernel exist only in -DMI355_QMM_PROBES builds (tools/): every guarded load or branch in the hot
    // loop costs the production kernel measurable time
    const int dbg = QMM_DBG(a);
    constexpr int NV = BT < 4 ? BT : 4;
    constexpr int PF = (R > QMM_PF_MIN) ? R : QMM_PF_MIN;
    constexpr int PFK = PF / R;                                  // ring depth in k-blocks
    static_assert(PF % R == 0, "ring depth must be a multiple of the tiles per workgroup");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
#ifndef QMM_UNIFORM_WAVE
#define QMM_UNIFORM_WAVE 1
#endif
#if QMM_UNIFORM_WAVE   // the wave index as a SCALAR: the k-block loop and its `active` tests become scalar branches instead of exec-masked regions
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), NW = blockDim.x >> 6;
#else
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, NW = blockDim.x >> 6;
#endif
    const int nkb = a.K >> 8;

    // ---- which tiles does this workgroup own?  slot r in [0,R)
    int segi[R], tile[R];
    if (a.paired) {
#pragma unroll
        for (int r = 0; r < R; ++r) { segi[r] = r & 1; tile[r] = blockIdx.x * (R / 2 > 0 ? R / 2 : 1) + (r >> 1); }
    } else {
        // (the segment walk over values that are already in SGPRs: a loop over a.seg[s] re-loads from the kernarg segment with a
        // run-time index, one dependent scalar round trip per segment in front of the first weight request)
        int t = blockIdx.x * R, s = 0;
#if QMM_SEG_SELECT
        const int nt0 = a.seg[0].n_tiles, nt1 = a.seg[1].n_tiles;
        if (a.nseg > 1 && t >= nt0) {
            t -= nt0; s = 1;
            if (a.nseg > 2 && t >= nt1) { t -= nt1; s = 2; }
        }
#else       // the round-1..5 form (A/B builds: -DQMM_KARG_BURST=0)
        while (s + 1 < a.nseg && t >= a.seg[s].n_tiles) { t -= a.seg[s].n_tiles; ++s; }
#endif
#pragma unroll
        for (int r = 0; r < R; ++r) { segi[r] = s; tile[r] = t + r; }      // launcher guarantees n_tiles % R == 0
    }
    const uint8_t* wbase[R];
    int wtype[R], wtb[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
#if QMM_SEG_SELECT
        // (seg_pick: opaque SGPR selects -- a plain ?: chain over the three segments is turned into a lookup table in SCRATCH by hipcc,
        // and the kernel then waits for a scratch load in front of its first weight request)
        wtype[r] = WT ? WT : seg_pick(a.seg[0].type, a.seg[1].type, a.seg[2].type, segi[r]);
        wtb[r] = (wtype[r] == MI355_GGML_Q4_K) ? Q4K_TILE : Q6K_TILE;
        {   // (the pointer keeps its type through the selects: rebuilt from integer halves it loses the GLOBAL address space and every
            // weight load becomes a flat load that counts against lgkmcnt as well -- no counted wait survives that)
            const uint8_t* wp = a.seg[0].w;
            if (segi[r] == 1) wp = a.seg[1].w;
            else if (segi[r] == 2) wp = a.seg[2].w;
            wbase[r] = wp + (size_t)tile[r] * nkb * wtb[r];
        }
#else
        wtype[r] = WT ? WT : a.seg[segi[r]].type;
        wtb[r] = (wtype[r] == MI355_GGML_Q4_K) ? Q4K_TILE : Q6K_TILE;
        wbase[r] = a.seg[segi[r]].w + (size_t)tile[r] * nkb * wtb[r];
#endif
        if constexpr (MOE) wbase[r] += (segi[r] == 0 ? po.w0 : segi[r] == 1 ? po.w1 : po.w2);
    }
    // RMSNorm weights, or any readable K floats when no norm is fused (keeps the load count static)
    const float* nwp = a.norm_w ? a.norm_w : reinterpret_cast<const float*>(a.seg[0].w);

    // ---- LDS carve-up: per-wave activation scratch, then the cross-wave reduction area
    constexpr int XW = 32 * 2 * BT * 16;                          // bytes per wave: the hi/lo fragment image of one k-block
    uint8_t* ximg = smem + (size_t)wave * XW;
    float* red = reinterpret_cast<float*>(smem + (size_t)NW * XW);   // [NW][R][BT][16]
    float* red_ss = red + (size_t)NW * R * BT * 16;               // [NW][BT]

    float y[R][NV];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int v = 0; v < NV; ++v) y[r][v] = 0.f;
    float ss[BT];
#pragma unroll
    for (int b = 0; b < BT; ++b) ss[b] = 0.f;

    EpiPre ep;
#ifdef QMM_NO_EPI_PRE      // bisect switch: no early operand fetch, the epilogue loads in place
    ep.pos = 0; ep.slot = -1; ep.f0 = 0.f; ep.f1 = 0.f; ep.lrow = 0; ep.sgi = 0; ep.live = false; ep.rope = false;
#else
    epi_pre_early<BT, R>(a, segi, tile, ep);
#endif

    // this wave's k-blocks: kb = wave + NW*kbi, kbi in [0, n_my_kb); ring slot s <-> (kbi0 + s/R, tile s%R)
    const int n_my_kb = (nkb > wave) ? (nkb - wave + NW - 1) / NW : 0;
    const int kb_last = n_my_kb > 0 ? wave + NW * (n_my_kb - 1) : 0;
    XRegs<BT> xr[PFK];
    TileRegs buf[PF];
#pragma unroll
    for (int q = 0; q < PFK; ++q) {
        const int kb = wave + NW * q;
        xr[q] = load_x<BT, XB, NRM>(a, kb <= kb_last ? kb : kb_last, lane, nwp, MOE ? po.x_bytes : 0);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const bool ok = q < n_my_kb;
            buf[q * R + r] = load_tile<WT>(wtype[r], ok ? wbase[r] + (size_t)kb * wtb[r] : wbase[r], ok ? lane : 0);
        }
    }

#ifndef QMM_LATE_AFTER_LOOP
#define QMM_LATE_AFTER_LOOP 1
#endif
#if !QMM_LATE_AFTER_LOOP
    epi_pre_late<BT>(a, ep);
#endif
    QMM_TL(2);

    for (int kbi0 = 0; kbi0 < n_my_kb; kbi0 += PFK) {
#pragma unroll
        for (int q = 0; q < PFK; ++q) {
            const int kbi = kbi0 + q;
            const bool active = kbi < n_my_kb;                    // wave-uniform
            if (active && (dbg < 3 || dbg == 7)) stage_kblock3<BT, XB, NRM>(a, xr[q], ximg, lane, ss);
            QMM_TL_ONCE(3);
            const int kbn = wave + NW * (kbi + PFK);
            if (dbg < 3 || dbg == 7) xr[q] = load_x<BT, XB, NRM>(a, kbn <= kb_last ? kbn : kb_last, lane, nwp, MOE ? po.x_bytes : 0);
            const bool ok = kbi + PFK < n_my_kb;
            if constexpr (WT != 0) {
                // every tile of the launch has one type: the R tiles of this k-block share C_in and the A fragments
                if (active) {
                    if (dbg >= 1 && dbg != 7) {
#pragma unroll
                        for (int r = 0; r < R; ++r) {
                            const TileRegs& t = buf[q * R + r];
                            y[r][0] += __uint_as_float((t.a.x ^ t.b.x ^ t.c.x ^ t.d.x ^ t.b.w ^ t.c.w) & 0x3FFFFFu);
                        }
                    } else {
                        // at most two tiles share C_in / the A fragments: four at once cost 248 VGPRs (2 waves/SIMD)
                        constexpr int RJ = (R < 2 || (R > 2 && WT == MI355_GGML_Q6_K)) ? 1 : 2;   // Q6_K x 4 tiles (lm_head): one at a time keeps 3 waves/SIMD
#pragma unroll
                        for (int r0 = 0; r0 < R; r0 += RJ) {
                            if (WT == MI355_GGML_Q4_K) compute3_q4k<BT, NV, RJ>(&buf[q * R + r0], ximg, lane, y + r0);
                            else compute3_q6k<BT, NV, RJ>(&buf[q * R + r0], ximg, lane, y + r0);
#ifdef MI355_QMM_TIMELINE
                            if (qtl[4] == 0) { asm volatile("" ::"v"(y[r0][0])); qtl[4] = wall_clock64(); }
#endif
                        }
                    }
                }
#pragma unroll
                for (int r = 0; r < R; ++r)
                    buf[q * R + r] = load_tile<WT>(wtype[r], ok ? wbase[r] + (size_t)kbn * wtb[r] : wbase[r], ok ? lane : 0);   // unconditional: a guarded load here costs the counted vmcnt waits (measured -7 % at batch 1)
            } else {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int s = q * R + r;
                    if (active) {
                        if (dbg >= 1 && dbg != 7) {
                            y[r][0] += __uint_as_float((buf[s].a.x ^ buf[s].b.x ^ buf[s].c.x ^ buf[s].d.x ^ buf[s].b.w ^ buf[s].c.w) & 0x3FFFFFu);
                        } else if (wtype[r] == MI355_GGML_Q4_K) {
                            compute3_q4k<BT, NV, 1>(&buf[s], ximg, lane, &y[r]);
                        } else {
                            compute3_q6k<BT, NV, 1>(&buf[s], ximg, lane, &y[r]);
                        }
#ifdef MI355_QMM_TIMELINE
                        if (qtl[4] == 0) { asm volatile("" ::"v"(y[r][0])); qtl[4] = wall_clock64(); }
#endif
                    }
                    buf[s] = load_tile<WT>(wtype[r], ok ? wbase[r] + (size_t)kbn * wtb[r] : wbase[r], ok ? lane : 0);
                }
            }
        }
    }

    if (dbg == 4) {                                              // probe: weight stream only, no reduction / epilogue
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < R; ++r) t += y[r][0];
        if (t == 1.2345e-30f) a.out[0] = t;
        return;
    }
    qmm_stamp(a, 1);
    QMM_TL(5);
#if QMM_LATE_AFTER_LOOP
    // cos / sin of the RoPE lanes are requested HERE, behind the k-block loop (round 6): asked for in the prologue they made the wave that holds
    // the epilogue lanes wait for the position's round trip before its first staging step -- the straggler of every q|k|v workgroup
    // (profiles/r06_b1_inside_launch.txt); here their latency sits under the cross-wave reduction and the barrier
    epi_pre_late<BT>(a, ep);
#endif
    // ---- hi + lo, then cross-wave reduction in LDS.  After the xor-32 add, lanes 0..31 hold batch 4*kg+v.
    const int kg = lane >> 4, row = lane & 15;
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            y[r][v] += __shfl_xor(y[r][v], 32, 64);
            const int b = 4 * kg + v;
            if (kg < 2 && b < BT) red[(((size_t)wave * R + r) * BT + b) * 16 + row] = y[r][v];
        }
    if (a.norm_w) {
#pragma unroll
        for (int b = 0; b < BT; ++b) {
            const float t = wave_sum(ss[b]);
            if (lane == 0) red_ss[wave * BT + b] = t;
        }
    }
    __syncthreads();
    qmm_stamp(a, 2);
    QMM_TL(6);

    qmm_epilogue<BT, R>(a, red, red_ss, NW, segi, tile, ep, MOE ? po.out : 0);
    qmm_stamp(a, 3);
    QMM_TL(7);
    QMM_TL_FLUSH();
}

template <int BT, int R, int WT, int XB, int NRM>
__global__ void __launch_bounds__(512) qmm_kernel(const QmmArgs a) { qmm_body<BT, R, WT, XB, NRM>(a); }
#include "qmm_exact.inc"   // parity mode (tests only): exact single-token mat-vecs behind the same fused epilogue
#ifdef MI355_QMM_PROBES   // probe builds only (tools/build_probe_lib.sh): the Q8_K-activation experiment (round 2)
#include "probes/qmm_q8.inc"
#endif

// MoE variant (decode-shaped, BT = 1): blockIdx.y = (token, slot) pair.  The adjusted descriptor is a private copy
// -- kept out of the dense kernel, where the kernel arguments must stay scalar loads from the kernarg segment.
template <int R, int WT, int XB, int NRM>
__global__ void __launch_bounds__(512) qmm_moe_kernel(const QmmArgs a) {
    const int pair = blockIdx.y;
    const size_t e = (size_t)a.moe_expert[pair];
    const size_t xes = (a.x_dtype == MI355_DTYPE_BF16) ? 2 : 4;
    const QmmPairOff po = {e * (size_t)a.moe_stride[0], e * (size_t)a.moe_stride[1], e * (size_t)a.moe_stride[2],
                           (size_t)(pair / a.moe_xdiv) * a.ldx * xes, (size_t)pair * a.ldo};
    qmm_body<1, R, WT, XB, NRM, true>(a, po);
}

// ================================================================================================
// Wide path (9..32 tokens per launch): every weight byte is still read once.  GEMM-style: the consumer waves of a
// workgroup own different row tiles and sweep K together, sharing one LDS image of the activations per k-block
// (double-buffered, one barrier per k-block).  The image (f16 hi / lo fragments of x / 16, lo scaled by 2^11, + the bf16
// pieces of the 32-element sub-block sums; RMSNorm weight already applied) is produced by `qmg_prep_entry` -- in the staging launch or in the previous mat-mul's epilogue.
//   image of k-block kb:  ximg [32 entries][MT*16 rows][16 B] | S32 fragments [MT][4][16][4 bf16]   (1088 B per token row)
//   rows of M-tile mt: mt*16 + m, m<8 = hi(batch 8mt+m), m>=8 = lo(batch 8mt+m-8)
#define QMW_MAXMT 4

// The Q4_K minimum term is NOT applied per sub-block in VALU but by one K=16 MFMA per m-tile against the sub-block sums
// staged as an A fragment (bf16 hi / lo pieces of S):  T1 = sum_j m_j S_j  ->  y -= dmin * T1.  (The second generation also
// needed T2 = sum_j sc_j S_j for the "+128" of its bf16 code trick; the f16 operands of the third carry no offset.)
__device__ __forceinline__ uint2 bytes4_to_bf16x4(uint32_t w) {
    // integers < 256 are exact in bf16 = the upper half of their f32: plain bit operations.  (Not cvt_pk_bf16: that is
    // an inline-asm VALU write, and hipcc pads no wait states between an asm-written VGPR and an MFMA that reads it as an
    // operand right away -- measured as wrong sums on the first m-tile only.)
    const float f0 = (float)(w & 0xFF), f1 = (float)((w >> 8) & 0xFF), f2 = (float)((w >> 16) & 0xFF), f3 = (float)(w >> 24);
    return make_uint2((__float_as_uint(f0) >> 16) | (__float_as_uint(f1) & 0xFFFF0000u),
                      (__float_as_uint(f2) >> 16) | (__float_as_uint(f3) & 0xFFFF0000u));
}
// Third-generation wide operands: the image is f16 (hi = f16(x / 16), lo = f16(x / 16 - hi): 22 mantissa bits, range
// 1e6) so that the B operand can carry the 6-bit sub-block scale EXACTLY: f16(1024 + q) is `0x6400 | q`, one packed
// fma(1024 + q, sc, -1024 sc) = sc q (<= 945 < 2048: exact).  The eight sub-blocks of a k-block then chain in the MFMA
// accumulator and the super-block scale is applied once:  y += 16 d acc - 16 dmin sum_j m_j S_j.  Per 2304-B unit at 32
// tokens the VALU work drops from ~300 to ~170 instructions (the launch was VALU-issue-bound: 8.6 M VALU instructions per
// gate/up launch = 14 us of a SIMD's issue slots out of 32 us, with the scale FMAs chained behind every MFMA).
typedef _Float16 qmg_h2 __attribute__((ext_vector_type(2)));
typedef _Float16 qmg_h4 __attribute__((ext_vector_type(4)));
typedef _Float16 qmg_h8 __attribute__((ext_vector_type(8)));
#define QMG_XSCALE 0.0625f          // image = x / 16 ...
#define QMG_XUNSCALE 16.0f          // ... folded back into d / dmin
#define QMG_LOSCALE 2048.0f         // lo plane = (x / 16 - hi) * 2^11 ...
#define QMG_LOUNSCALE (1.0f / 2048.0f)   // ... folded back when the hi and lo rows of the result meet
template <int MT>
__device__ __forceinline__ void wide_q4k2(const TileRegs& w, const uint8_t* __restrict__ L, int lane, float (&y)[MT][4]) {
    const int m = lane & 15, kg = lane >> 4;
    const uint8_t* sfrag = L + (size_t)32 * MT * 16 * 16;           // [MT][kg 4][row 16][4 bf16]
    const float d = QMG_XUNSCALE * f16_bits_to_f32((uint16_t)(w.a.x & 0xFFFF));
    const float ndmin = -QMG_XUNSCALE * f16_bits_to_f32((uint16_t)(w.a.x >> 16));
    const uint32_t s0 = w.a.y, s1 = w.a.z, s2 = w.a.w;
    const uint32_t scl = s0 & 0x3F3F3F3Fu, mnl = s1 & 0x3F3F3F3Fu;
    const uint32_t sch = (s2 & 0x0F0F0F0Fu) | ((s0 >> 2) & 0x30303030u);
    const uint32_t mnh = ((s2 >> 4) & 0x0F0F0F0Fu) | ((s1 >> 2) & 0x30303030u);
    // B fragment of the minimum term: k = 4kg + e  <->  j = 4(kg & 1) + e  (both pieces of S use the same values)
    const uint2 mb = bytes4_to_bf16x4((kg & 1) ? mnh : mnl);
    uint32_t nib = 0x000F000Fu, c1024 = 0x64006400u;
    asm volatile("" : "+v"(nib), "+v"(c1024));                       // in VGPRs: "(w >> s) & nib | c1024" is one v_and_or_b32
    const qmg_h2 k1024 = {(_Float16)1024.f, (_Float16)1024.f}, kneg = {(_Float16)(-1024.f), (_Float16)(-1024.f)};
    // the eight scales as f16 pairs: 1024 + sc is 0x6400 | sc, minus 1024 in one packed add
    qmg_h2 scp[4];                                                    // (sc0,sc2) (sc1,sc3) (sc4,sc6) (sc5,sc7)
    scp[0] = __builtin_bit_cast(qmg_h2, (scl & 0x00FF00FFu) | c1024) - k1024;
    scp[1] = __builtin_bit_cast(qmg_h2, ((scl >> 8) & 0x00FF00FFu) | c1024) - k1024;
    scp[2] = __builtin_bit_cast(qmg_h2, (sch & 0x00FF00FFu) | c1024) - k1024;
    scp[3] = __builtin_bit_cast(qmg_h2, ((sch >> 8) & 0x00FF00FFu) | c1024) - k1024;
    f32x4_t acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const qmg_h2 pr2 = scp[2 * (j >> 2) + (j & 1)];               // holds sc_j in half (j >> 1) & 1
        const _Float16 sj = ((j >> 1) & 1) ? pr2[1] : pr2[0];
        const qmg_h2 S = {sj, sj}, OFF = S * kneg;
        const int p = j >> 2, pr = (j >> 1) & 1, sh = (j & 1) * 4;
        const uint4 qs = p ? w.c : w.b;
        const uint32_t w0 = pr ? qs.z : qs.x, w1 = pr ? qs.w : qs.y;
        const qmg_h2 r0 = __builtin_elementwise_fma(__builtin_bit_cast(qmg_h2, ((w0 >> sh) & nib) | c1024), S, OFF);
        const qmg_h2 r1 = __builtin_elementwise_fma(__builtin_bit_cast(qmg_h2, ((w0 >> (sh + 8)) & nib) | c1024), S, OFF);
        const qmg_h2 r2 = __builtin_elementwise_fma(__builtin_bit_cast(qmg_h2, ((w1 >> sh) & nib) | c1024), S, OFF);
        const qmg_h2 r3 = __builtin_elementwise_fma(__builtin_bit_cast(qmg_h2, ((w1 >> (sh + 8)) & nib) | c1024), S, OFF);
        const qmg_h8 bw = {r0[0], r0[1], r1[0], r1[1], r2[0], r2[1], r3[0], r3[1]};
        const uint8_t* abase = L + ((size_t)(4 * j + kg) * (MT * 16) + m) * 16;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const uint4 aw = *reinterpret_cast<const uint4*>(abase + (size_t)mt * 16 * 16);
            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(qmg_h8, aw), bw, acc[mt], 0, 0, 0);
        }
    }
    const f32x4_t zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const uint2 sa = *reinterpret_cast<const uint2*>(sfrag + ((size_t)(mt * 4 + kg) * 16 + m) * 8);
        const f32x4_t t1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4_t, sa), __builtin_bit_cast(s16x4_t, mb), zero, 0, 0, 0);
#pragma unroll
        for (int v = 0; v < 4; ++v) y[mt][v] = fmaf(ndmin, t1[v], fmaf(d, acc[mt][v], y[mt][v]));
    }
}

// Q6_K on the wide path.  The int8 sub-block scale times the 6-bit code (|sc (c - 32)| <= 4064) does not fit f16's 11
// bits, so the scale is split sc = 2 sh + sl (sh = sc >> 1, sl = sc & 1): |sh (c - 32)| <= 2048 and sl (c - 32) are exact.
// The code rides in mantissa bits 9..4 of f16 64.0 (`0x5400 | c << 4` = 64 + c): one packed fma(64 + c, sh, -96 sh) =
// sh (c - 32) per part, two MFMA chains per k-block (hi part, lo part) and ONE scaling at the end, y += 16 d (2 acc_h +
// acc_l) -- instead of a scale FMA per sub-block, m-tile and output (469 -> ~280 VALU per 3360-B tile at 32 tokens; the
// MFMA count doubles, the matrix pipe was 23 % busy).  A K=32 MFMA spans the two 16-element sub-blocks 2p, 2p+1.
template <int MT>
__device__ __forceinline__ void wide_q6k(const TileRegs& w, const uint8_t* __restrict__ L, int lane, float (&y)[MT][4]) {
    const int m = lane & 15, kg = lane >> 4;
    const float d = QMG_XUNSCALE * f16_bits_to_f32((uint16_t)(w.e & 0xFFFF));
    const uint32_t scw[4] = {w.a.x, w.a.y, w.a.z, w.a.w};
    const uint32_t qhw[4] = {w.d.x, w.d.y, w.d.z, w.d.w};
    uint32_t cmask = 0x03F003F0u, c64 = 0x54005400u, c1024 = 0x64006400u, bmask = 0x00FF00FFu;
    asm volatile("" : "+v"(cmask), "+v"(c64), "+v"(c1024), "+v"(bmask));
    const qmg_h2 k1088 = {(_Float16)1088.f, (_Float16)1088.f}, k1024 = {(_Float16)1024.f, (_Float16)1024.f};
    const qmg_h2 kn96 = {(_Float16)(-96.f), (_Float16)(-96.f)};
    // scale parts of the 16 sub-blocks as f16 pairs: word q holds sub-blocks 4q..4q+3; [q][0] = (s0, s2), [q][1] = (s1, s3)
    qmg_h2 SH[4][2], SL[4][2];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t uh = ((scw[q] ^ 0x80808080u) >> 1) & 0x7F7F7F7Fu;      // bytes sh + 64
        const uint32_t ul = scw[q] & 0x01010101u;                              // bytes sl
        SH[q][0] = __builtin_bit_cast(qmg_h2, (uh & bmask) | c1024) - k1088;
        SH[q][1] = __builtin_bit_cast(qmg_h2, ((uh >> 8) & bmask) | c1024) - k1088;
        SL[q][0] = __builtin_bit_cast(qmg_h2, (ul & bmask) | c1024) - k1024;
        SL[q][1] = __builtin_bit_cast(qmg_h2, ((ul >> 8) & bmask) | c1024) - k1024;
    }
    f32x4_t acch[MT], accl[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) { acch[mt] = f32x4_t{0.f, 0.f, 0.f, 0.f}; accl[mt] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const uint4 ql = n ? w.c : w.b;
        uint32_t t[2][4];
#pragma unroll
        for (int is = 0; is < 2; ++is) {
            const uint32_t a = is ? ql.z : ql.x, b = is ? ql.w : ql.y, h = qhw[2 * n + is];
            t[is][0] = (a & 0x0F0F0F0Fu) | ((h << 4) & 0x30303030u);
            t[is][1] = (b & 0x0F0F0F0Fu) | ((h << 2) & 0x30303030u);
            t[is][2] = ((a >> 4) & 0x0F0F0F0Fu) | (h & 0x30303030u);
            t[is][3] = ((b >> 4) & 0x0F0F0F0Fu) | ((h >> 2) & 0x30303030u);
        }
#pragma unroll
        for (int tt = 0; tt < 4; ++tt) {
            qmg_h2 bh[4], bl[4];
#pragma unroll
            for (int is = 0; is < 2; ++is) {
                const int s = 8 * n + 2 * tt + is;                    // 16-element sub-block 0..15
                const qmg_h2 ph = SH[s >> 2][s & 1], pl = SL[s >> 2][s & 1];
                const _Float16 sh = ((s >> 1) & 1) ? ph[1] : ph[0], sl = ((s >> 1) & 1) ? pl[1] : pl[0];
                const qmg_h2 Sh = {sh, sh}, Sl = {sl, sl}, Oh = Sh * kn96, Ol = Sl * kn96;
                const qmg_h2 c02 = __builtin_bit_cast(qmg_h2, ((t[is][tt] << 4) & cmask) | c64);     // elements (b0, b2): 64 + c
                const qmg_h2 c13 = __builtin_bit_cast(qmg_h2, ((t[is][tt] >> 4) & cmask) | c64);     // elements (b1, b3)
                bh[2 * is] = __builtin_elementwise_fma(c02, Sh, Oh); bh[2 * is + 1] = __builtin_elementwise_fma(c13, Sh, Oh);
                bl[2 * is] = __builtin_elementwise_fma(c02, Sl, Ol); bl[2 * is + 1] = __builtin_elementwise_fma(c13, Sl, Ol);
            }
            const qmg_h8 Bh = {bh[0][0], bh[0][1], bh[1][0], bh[1][1], bh[2][0], bh[2][1], bh[3][0], bh[3][1]};
            const qmg_h8 Bl = {bl[0][0], bl[0][1], bl[1][0], bl[1][1], bl[2][0], bl[2][1], bl[3][0], bl[3][1]};
            // A: elements 4kg..4kg+3 of sub-block 2p (entry 2 s0 + (kg >> 1), half kg & 1) and the same of sub-block 2p + 1
            const int s0 = 8 * n + 2 * tt;
            const uint8_t* abase = L + ((size_t)(2 * s0 + (kg >> 1)) * (MT * 16) + m) * 16 + (kg & 1) * 8;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const uint2 a0 = *reinterpret_cast<const uint2*>(abase + (size_t)mt * 16 * 16);
                const uint2 a1 = *reinterpret_cast<const uint2*>(abase + (size_t)mt * 16 * 16 + (size_t)2 * (MT * 16) * 16);
                const qmg_h8 A = __builtin_bit_cast(qmg_h8, make_uint4(a0.x, a0.y, a1.x, a1.y));
                acch[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, Bh, acch[mt], 0, 0, 0);
                accl[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A, Bl, accl[mt], 0, 0, 0);
            }
        }
    }
    const float d2 = 2.f * d;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int v = 0; v < 4; ++v) y[mt][v] = fmaf(d2, acch[mt][v], fmaf(d, accl[mt][v], y[mt][v]));
}

// ================================================================================================
// GEMM and epilogue are split.  (A first generation kept one wave on a row tile for ALL of K with the epilogue fused,
// which capped the grid at tiles/NW workgroups and ran 1-2 waves per SIMD at ~200 VGPRs: latency-bound, 0.9 TB/s on
// gate/up at 32 tokens; removed.)  The GEMM kernel splits K across
// gridDim.y so that every launch has >= ~2048 waves, stays under 128 VGPRs (4 waves per SIMD, two 8-wave workgroups
// per CU), stages the activation image global -> LDS by DMA (no staging registers), and writes f32 partial sums
// [ks][token][row]; a small second kernel adds the partials and applies the epilogue (deterministic: no atomics).
static inline size_t qmg_kb_bytes(int MT) { return (((size_t)MT * 8 * 1088) + 1023) / 1024 * 1024; }


// image + per-k-block sum of squares in ONE pass: grid = k-blocks, a workgroup builds k-block kb for every token
// row.  The RMSNorm weight is applied here, the 1/rms factor (a per-token scalar) is applied by the epilogue kernel
// from the per-k-block partial sums ssp[kb][row] -- no second pass over x, no atomics.
// one image entry (8 consecutive elements El of k-block kb, token row b): f16 hi / lo fragments, sub-block sums, and the
// sum of squares of the row's k-block (reduced over the 32 consecutive lanes that hold its 32 entries)
__device__ __forceinline__ void qmg_prep_entry(uint8_t* __restrict__ img, float* __restrict__ ssp, float (&v)[8], const bool live,
                                               const float* __restrict__ norm_w, const int MT, const size_t kbb,
                                               const int kb, const int b, const int El) {
    // the two call sites (staging kernel, chained epilogue) must produce the same bits from the same values: no
    // compiler-chosen fused multiply-adds in the sums below (a contraction picked differently per kernel = 1 ulp of 1/rms)
#pragma clang fp contract(off)
    const int BP = MT * 8;
    uint8_t* kbase = img + (size_t)kb * kbb;
    const int mt = b >> 3, m = b & 7;
    const int k = kb * 256 + El * 8;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) ss += v[i] * v[i];
    if (norm_w && live) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] *= norm_w[k + i];
    }
    // f16 hi / lo of x / 16 (element order {0,2,1,3,4,6,5,7} inside the 8-group, as the B operands unpack)
    auto pk = [](float a0, float a1) {
        const _Float16 h0 = (_Float16)a0, h1 = (_Float16)a1;
        return (uint32_t)__builtin_bit_cast(uint16_t, h0) | ((uint32_t)__builtin_bit_cast(uint16_t, h1) << 16);
    };
    auto lo16 = [](uint32_t w2) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(w2 & 0xFFFF)); };
    auto hi16 = [](uint32_t w2) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(w2 >> 16)); };
    float xs[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) xs[i] = v[i] * QMG_XSCALE;       // |x| >= 1.05e6 leaves f16: inf on both planes = NaN results, never a plausible number
    // no f16 denormals on either plane: a value below f16's smallest normal goes to the (scaled) lo plane entirely
    auto nz = [](float t) { return fabsf(t) < 6.103515625e-05f ? 0.f : t; };
    const uint32_t h02 = pk(nz(xs[0]), nz(xs[2])), h13 = pk(nz(xs[1]), nz(xs[3]));
    const uint32_t h46 = pk(nz(xs[4]), nz(xs[6])), h57 = pk(nz(xs[5]), nz(xs[7]));
    const float hf[8] = {lo16(h02), lo16(h13), hi16(h02), hi16(h13), lo16(h46), lo16(h57), hi16(h46), hi16(h57)};
    // the lo plane is stored times 2^11 (the residual of an f16 rounding is 2^-11 of the value: unscaled it would sit in
    // f16's denormal range for every |x| < 2); the GEMM folds 2^-11 back into the lo rows of its result
    const uint32_t l02 = pk((xs[0] - hf[0]) * QMG_LOSCALE, (xs[2] - hf[2]) * QMG_LOSCALE), l13 = pk((xs[1] - hf[1]) * QMG_LOSCALE, (xs[3] - hf[3]) * QMG_LOSCALE);
    const uint32_t l46 = pk((xs[4] - hf[4]) * QMG_LOSCALE, (xs[6] - hf[6]) * QMG_LOSCALE), l57 = pk((xs[5] - hf[5]) * QMG_LOSCALE, (xs[7] - hf[7]) * QMG_LOSCALE);
    float hsum = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) hsum += hf[i];
    const float lsum = (lo16(l02) + hi16(l02)) + (lo16(l13) + hi16(l13)) + (lo16(l46) + hi16(l46)) + (lo16(l57) + hi16(l57));
    uint8_t* ent = kbase + ((size_t)El * (MT * 16) + mt * 16) * 16;
    *reinterpret_cast<uint4*>(ent + (size_t)m * 16) = make_uint4(h02, h13, h46, h57);
    *reinterpret_cast<uint4*>(ent + (size_t)(8 + m) * 16) = make_uint4(l02, l13, l46, l57);
    const float h16 = hsum + __shfl_xor(hsum, 1, 64), l16 = lsum + __shfl_xor(lsum, 1, 64);
    const float h32 = h16 + __shfl_xor(h16, 2, 64), l32 = l16 + __shfl_xor(l16, 2, 64);       // 32-element sub-block sums
    if ((El & 3) == 0) {
        // sub-block sums as the A operand of a K=16 MFMA (Q4_K minimum / offset terms, wide_q4k2): per m-tile
        // [kg 4][row 16][4 bf16], k = 8*piece + j with piece 0 = bf16(S), piece 1 = bf16(S - piece 0) (S to 2^-17, like x);
        // rows 0..7 = sums of the hi plane of the tile's 8 tokens, rows 8..15 = sums of the lo plane
        const int j = El >> 2;
        uint8_t* sfrag = kbase + (size_t)32 * MT * 16 * 16 + (size_t)mt * 512;
        const uint16_t hh = f32_to_bf16(h32), lh = f32_to_bf16(l32);
        const uint16_t hl = f32_to_bf16(h32 - bf16_to_f32(hh)), ll = f32_to_bf16(l32 - bf16_to_f32(lh));
        auto at = [&](int piece, int row) { return reinterpret_cast<uint16_t*>(sfrag + ((size_t)(2 * piece + (j >> 2)) * 16 + row) * 8) + (j & 3); };
        *at(0, m) = hh; *at(1, m) = hl;
        *at(0, 8 + m) = lh; *at(1, 8 + m) = ll;
    }
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) ss += __shfl_xor(ss, o, 64);
    if (El == 0) ssp[(size_t)kb * BP + b] = ss;
}

template <int MT>
__global__ void __launch_bounds__(256) qmm_prep2_kernel(uint8_t* __restrict__ img, float* __restrict__ ssp, const QmmArgs a, const size_t kbb) {
    constexpr int BP = MT * 8;
    const int kb = blockIdx.x;
    // grid = (k-blocks, m-tiles): one workgroup = the 8 token rows of one m-tile (a single pass: the launch is a chain
    // of one load and one store latency, so it must not loop)
    for (int e = blockIdx.y * blockDim.x + threadIdx.x; e < BP * 32; e += blockDim.x * gridDim.y) {   // 32 consecutive lanes = the 32 entries of one row
        const int b = e >> 5, El = e & 31;
        const bool live = b < a.B;
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = 0.f;
        const int k = kb * 256 + El * 8;
        if (live) {
            if (a.x_dtype == MI355_DTYPE_BF16) {
                const uint4 w = *reinterpret_cast<const uint4*>(static_cast<const uint16_t*>(a.x) + (size_t)b * a.ldx + k);
                v[0] = bf16lo_to_f32(w.x); v[1] = bf16hi_to_f32(w.x); v[2] = bf16lo_to_f32(w.y); v[3] = bf16hi_to_f32(w.y);
                v[4] = bf16lo_to_f32(w.z); v[5] = bf16hi_to_f32(w.z); v[6] = bf16lo_to_f32(w.w); v[7] = bf16hi_to_f32(w.w);
            } else {
                const float* xp = static_cast<const float*>(a.x) + (size_t)b * a.ldx + k;
                const float4 v0 = *reinterpret_cast<const float4*>(xp), v1 = *reinterpret_cast<const float4*>(xp + 4);
                v[0] = v0.x; v[1] = v0.y; v[2] = v0.z; v[3] = v0.w; v[4] = v1.x; v[5] = v1.y; v[6] = v1.z; v[7] = v1.w;
            }
        }
        qmg_prep_entry(img, ssp, v, live, a.norm_w, MT, kbb, kb, b, El);
    }
}

// Wave specialisation: waves 0..6 are CONSUMERS (one row tile each: weight stream with a two-deep register prefetch +
// unpack + MFMA), wave 7 is the LOADER of the shared activation image (global -> LDS DMA for the next k-block).  Keeping
// the LDS-DMA out of the consumers' instruction stream matters: with a DMA in flight hipcc puts `s_waitcnt vmcnt(0)` in
// front of the next use of ANY loaded register, which serialised load -> wait -> compute in the first version
// (measured: 36 of 45 us of the gate/up GEMM were the load/barrier skeleton).  One barrier per k-block.
#define QMG_NC 7
// segments [s0, s1) of `a` (all of type WT when WT != 0) = row-tile slots [0, n_slots) of this run, workgroup bx of the run;
// NC consumer waves + one loader wave per workgroup
template <int MT, int WT, int NC = QMG_NC>
__device__ __forceinline__ void qmm_gemm_body(const QmmArgs& a, const uint8_t* __restrict__ img, float* __restrict__ part, const int ldp,
                                              const int s0, const int s1, const int n_slots, const int slot_base, const int bx) {
    extern __shared__ __attribute__((aligned(1024))) uint8_t smem[];
    constexpr int BP = MT * 8;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nkb = a.K >> 8;
    const size_t kbb = (((size_t)MT * 8 * 1088) + 1023) / 1024 * 1024;
    const int kb_per = (nkb + gridDim.y - 1) / gridDim.y;
    const int kb_lo = blockIdx.y * kb_per, kb_hi = min(nkb, kb_lo + kb_per);
    if (kb_lo >= kb_hi) return;                                       // uniform for the workgroup

    if (wave == NC) {
        // ---------------- loader wave: the image of k-block kb+1 lands while the consumers work on kb
        const int nchunk = (int)(kbb >> 10);                          // 1 KiB = one wave-wide 16-B DMA
        auto dma_kb = [&](int kb, int buf) {
            const uint8_t* src = img + (size_t)kb * kbb;
            uint8_t* dst = smem + (size_t)buf * kbb;
            for (int c = 0; c < nchunk; ++c)
                __builtin_amdgcn_global_load_lds((qmg_gptr_t)(src + (size_t)c * 1024 + lane * 16), (qmg_lptr_t)(dst + (size_t)c * 1024), 16, 0, 0);
        };
        dma_kb(kb_lo, kb_lo & 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int kb = kb_lo; kb < kb_hi; ++kb) {
            if (kb + 1 < kb_hi && QMM_DBG(a) != 2) dma_kb(kb + 1, (kb + 1) & 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        return;
    }
    // ---------------- consumer waves
    int slot = bx * NC + wave;
    const bool have = slot < n_slots;
    if (!have) slot = 0;
    int t = slot, sg = s0;
    while (sg + 1 < s1 && t >= a.seg[sg].n_tiles) { t -= a.seg[sg].n_tiles; ++sg; }
    const int wtype = WT ? WT : a.seg[sg].type;                       // WT != 0: every tile of the launch has that type
    const int wtb = (wtype == MI355_GGML_Q4_K) ? Q4K_TILE : Q6K_TILE;
    const uint8_t* wbase = a.seg[sg].w + (size_t)t * nkb * wtb;
    float y[1][MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int v = 0; v < 4; ++v) y[0][mt][v] = 0.f;
    TileRegs cur = load_tile<WT>(wtype, wbase + (size_t)kb_lo * wtb, lane);
    __syncthreads();                                                  // first image landed
    for (int kb = kb_lo; kb < kb_hi; ++kb) {
        const uint8_t* Lc = smem + (size_t)(kb & 1) * kbb;
        const bool more = kb + 1 < kb_hi;
        // the next tile streams in while this one is unpacked and multiplied (counted vmcnt: no DMA in this wave)
        TileRegs nxt = load_tile<WT>(wtype, more ? wbase + (size_t)(kb + 1) * wtb : wbase, more ? lane : 0);
        if (QMM_DBG(a) == 1) { y[0][0][0] += __uint_as_float(cur.a.x ^ cur.b.y ^ cur.c.z); }
        else if (wtype == MI355_GGML_Q4_K) wide_q4k2<MT>(cur, Lc, lane, y[0]);
        else wide_q6k<MT>(cur, Lc, lane, y[0]);
        cur = nxt;
        __syncthreads();                                              // everyone is done with Lc; the loader filled the other buffer
    }
    // hi + lo halves of the M tile, then plain stores of the partial sums
    const int kg = lane >> 4, rr = lane & 15;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const float yy = kg >= 2 ? y[0][mt][v] * QMG_LOUNSCALE : y[0][mt][v];     // MFMA rows 8..15 = the lo plane
            y[0][mt][v] = yy + __shfl_xor(yy, 32, 64);
        }
    if (have && kg < 2) {
        float* pp = part + (size_t)blockIdx.y * BP * ldp + (size_t)(slot_base + slot) * 16 + rr;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int v = 0; v < 4; ++v) pp[(size_t)(8 * mt + 4 * kg + v) * ldp] = y[0][mt][v];
    }
}
template <int MT, int WT>
__global__ void __launch_bounds__(512, 4) qmm_gemm_kernel(const QmmArgs a, const uint8_t* __restrict__ img, float* __restrict__ part,
                                                          const int ldp, const int n_slots, const int slot_base) {
    qmm_gemm_body<MT, WT>(a, img, part, ldp, 0, a.nseg, n_slots, slot_base, (int)blockIdx.x);
}
// 15 consumers + the loader: twice the waves on one shared image per CU (the image DMA is 2.4x the weight bytes, so a
// second workgroup per CU is not the way to more waves).  Measured +7 % on gate/up only; kept as a tuning option.
#define QMG_NC_WIDE 15
template <int MT, int WT>
__global__ void __launch_bounds__(1024) qmm_gemm16_kernel(const QmmArgs a, const uint8_t* __restrict__ img, float* __restrict__ part,
                                                          const int ldp, const int n_slots, const int slot_base) {
    qmm_gemm_body<MT, WT, QMG_NC_WIDE>(a, img, part, ldp, 0, a.nseg, n_slots, slot_base, (int)blockIdx.x);
}
// two runs of segments in ONE launch (q, k in Q4_K + v in Q6_K: the v run alone is 10 workgroups x 13 us): the first nwg0
// workgroups take run 0 with the WTA body, the rest run 1 with the WTB body -- a uniform branch between two specialised
// bodies, not the per-wave type dispatch of a mixed build.  (At the 128-VGPR bound of the single-run kernels the two inlined
// bodies spill 39 dwords and the launch is as slow as the two it replaces; such launches have < 256 workgroups, so the
// second workgroup per CU that bound buys is never used.)
template <int MT, int WTA, int WTB>
__global__ void __launch_bounds__(512, 2) qmm_gemm2_kernel(const QmmArgs a, const uint8_t* __restrict__ img, float* __restrict__ part,
                                                           const int ldp, const int s_split, const int slots0, const int slots1) {
    const int nwg0 = (slots0 + QMG_NC - 1) / QMG_NC;
    if ((int)blockIdx.x < nwg0) qmm_gemm_body<MT, WTA>(a, img, part, ldp, 0, s_split, slots0, 0, (int)blockIdx.x);
    else qmm_gemm_body<MT, WTB>(a, img, part, ldp, s_split, a.nseg, slots1, slots0, (int)blockIdx.x - nwg0);
}

#include "qmm_wide1.inc"

// partial sums -> epilogue; one thread per (token, concatenated padded row).  Two phases, so that every load that does not
// depend on the deferred 1/rms factor (the k-split partial sums of the thread's row and of its partner row, the residual,
// the bias) is in flight BEFORE the workgroup reduces the sum of squares -- one memory round trip instead of two.
struct QmgEpiRow { int sg, lrow, orow; bool live, has_aux; float s_main, s_aux, b_main, b_aux, resid; };
__device__ __forceinline__ QmgEpiRow qmm_epilogue_load(const QmmArgs& a, const float* __restrict__ part, const int ldp, const int ks,
                                                       const int BP, const int prow, const int b) {
    QmgEpiRow e;
    e.sg = 0; e.lrow = prow;
    while (e.sg + 1 < a.nseg && e.lrow >= a.seg[e.sg].n_tiles * 16) { e.lrow -= a.seg[e.sg].n_tiles * 16; ++e.sg; }
    e.live = e.lrow < a.seg[e.sg].n_rows;
    e.orow = a.seg[e.sg].row0 + e.lrow;
    e.has_aux = false; e.s_main = e.s_aux = e.b_main = e.b_aux = e.resid = 0.f;
    if (!e.live) return e;
    int aux = 0, aux_orow = 0;
    if (a.epi == MI355_EPI_SILU_MUL) {                                // gate rows drive; up = same row of segment 1
        if (e.sg != 0) { e.live = false; return e; }
        e.has_aux = true; aux = a.seg[0].n_tiles * 16 + e.lrow; aux_orow = a.seg[1].row0 + e.lrow;
    } else if (a.epi == MI355_EPI_QKV_ROPE_CACHE) {
        if (e.sg < 2 && (e.lrow % a.D) < a.rot) { e.has_aux = true; aux = prow ^ 1; aux_orow = e.orow ^ 1; }
    }
    for (int k = 0; k < ks; ++k) {
        e.s_main += part[((size_t)k * BP + b) * ldp + prow];
        if (e.has_aux) e.s_aux += part[((size_t)k * BP + b) * ldp + aux];
    }
    if (a.bias) { e.b_main = a.bias[e.orow]; if (e.has_aux) e.b_aux = a.bias[aux_orow]; }
    if (a.epi == MI355_EPI_RESID) e.resid = a.resid[(size_t)b * a.ldo + e.orow];
    return e;
}
// q -> bf16 row of q_out; k / v -> bf16 (or e4m3) into the paged / flash cache slot of token b (EPI_QKV_ROPE_CACHE: segment 0 = q, 1 = k, 2 = v)
__device__ __forceinline__ void qmm_qkv_store(const QmmArgs& a, const int sg, const int lrow, const int b, const float o) {
    const int D = a.D, d = lrow % D, hh = lrow / D;
    const uint16_t ob = f32_to_bf16(o);
    if (sg == 0) {
        a.q_out[(size_t)b * a.Hq * D + lrow] = ob;
        return;
    }
    const int64_t slot = a.slot_mapping[b];
    if (slot < 0) return;
    uint16_t* cache = (sg == 1) ? a.kcache : a.vcache;
    if (a.kv_layout == MI355_KV_PAGED_FP8) {                // e4m3fn of the bf16 value, K layout x = 16
        uint8_t* c8 = reinterpret_cast<uint8_t*>(cache);
        const int64_t blk = slot / a.block_size, off = slot % a.block_size;
        const uint8_t q8 = to_e4m3(bf16_to_f32(ob));
        if (sg == 1) c8[((((blk * a.Hkv + hh) * (D / 16) + d / 16) * a.block_size + off) * 16) + d % 16] = q8;
        else c8[((blk * a.Hkv + hh) * D + d) * (int64_t)a.block_size + off] = q8;
    } else if (a.kv_layout == MI355_KV_FLASH) {
        cache[(slot * a.Hkv + hh) * D + d] = ob;
    } else {
        const int64_t blk = slot / a.block_size, off = slot % a.block_size;
        if (sg == 1) cache[((((blk * a.Hkv + hh) * (D / 8) + d / 8) * a.block_size + off) * 8) + d % 8] = ob;
        else cache[((blk * a.Hkv + hh) * D + d) * (int64_t)a.block_size + off] = ob;
    }
}
// returns the f32 value written to a.out for (token b, out column = the chain's k index), 0 when nothing was written
// (ooff: elements between a.out and the output of the workgroup's group -- grouped launches; 0 otherwise)
__device__ __forceinline__ float qmm_epilogue_apply(const QmmArgs& a, const QmgEpiRow& e, const float inv, const int b, const size_t ooff = 0) {
    if (!e.live) return 0.f;
    const int sg = e.sg, lrow = e.lrow, orow = e.orow;
    const float val = e.s_main * inv + e.b_main;
    if (a.epi == MI355_EPI_STORE) {
        a.out[ooff + (size_t)b * a.ldo + orow] = val;
        return val;
    } else if (a.epi == MI355_EPI_RESID) {
        const float o = e.resid + val;
        a.out[(size_t)b * a.ldo + orow] = o;
        return o;
    } else if (a.epi == MI355_EPI_SILU_MUL) {
        const float up = e.s_aux * inv + e.b_aux;
        const float o = silu_f(val) * up;
        a.out[ooff + (size_t)b * a.ldo + lrow] = o;
        return o;
    } else if (a.epi == MI355_EPI_QKV_ROPE_CACHE) {
        const int D = a.D, d = lrow % D, hh = lrow / D;
        float o = val;
        if (e.has_aux) {
            const float partner = e.s_aux * inv + e.b_aux;
            const int64_t pos = a.positions[b];
            const float c = a.cos_t[pos * (a.rot >> 1) + (d >> 1)], sn = a.sin_t[pos * (a.rot >> 1) + (d >> 1)];
            o = (d & 1) ? (partner * sn + val * c) : (val * c - partner * sn);
        }
        qmm_qkv_store(a, sg, lrow, b, o);
    }
    return 0.f;
}

// chain target: where the epilogue stages the NEXT wide mat-mul's activation image (img == null: no chain)
struct QmgChainOut { uint8_t* img; float* ssp; const float* norm_w; int K, MT; size_t kbb; int sp; };   // sp: 1 = single f16 plane (qmm_wide1.inc), MT then counts 16-token tiles

// grid = (padded rows / 256, tokens).  With a chain target the y extent covers the PADDED token rows (MT*8) and every
// workgroup = (token b, 256 consecutive output columns) = one (row, k-block) of the next image: the outputs meet in LDS
// and 32 threads build the 32 entries.
__device__ __forceinline__ void qmm_epilogue_body(const QmmArgs& a, const float* __restrict__ part, const int ldp, const int ks, const int BP,
                                                  const float* __restrict__ ssp, const QmgChainOut& ch, const float* __restrict__ rscale,
                                                  const size_t ooff, const int nb) {
    // (nb: live token rows -- a.B, or the row count of the workgroup's group; rows past it get zero entries in a chained image)
    const int prow = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    // deferred RMSNorm scale of this workgroup's token: the per-k-block partial sums are read by the lanes of one wave
    // in parallel (one L2 round trip) -- a per-thread loop over up to 56 k-blocks was most of this kernel's time
    __shared__ float sm_inv;
    float inv = 1.f;
    QmgEpiRow er;
    er.live = false;
    const bool mine = prow < ldp && b < nb;
    if (mine) er = qmm_epilogue_load(a, part, ldp, ks, BP, prow, b);
    if (a.norm_w && ssp) {
        if (threadIdx.x < 64) {
            float ss = 0.f;
            if (b < nb)
                for (int kb = threadIdx.x; kb < (a.K >> 8); kb += 64) ss += ssp[(size_t)kb * BP + b];
            ss = wave_sum(ss);
            if (threadIdx.x == 0) sm_inv = rsqrtf(ss / (float)a.K + a.eps);
        }
        __syncthreads();
        inv = sm_inv;
    }
    if (rscale && b < nb) inv *= rscale[b];                          // prompt-step GEMM: per-token power-of-two scale of the f16 image
    float o = 0.f;
    if (mine) o = qmm_epilogue_apply(a, er, inv, b, ooff);
    if (!ch.img) return;
    const int kb = blockIdx.x;
    if (kb * 256 >= ch.K) return;                                    // uniform: e.g. the `up` half of the gate/up rows
    __shared__ float sm_o[256];
    sm_o[threadIdx.x] = o;
    __syncthreads();
    if (threadIdx.x < 32) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = sm_o[threadIdx.x * 8 + i];
        if (ch.sp) qw1_prep_entry(ch.img, ch.ssp, v, b < nb, ch.norm_w, ch.MT, ch.kbb, kb, b, (int)threadIdx.x);
        else qmg_prep_entry(ch.img, ch.ssp, v, b < nb, ch.norm_w, ch.MT, ch.kbb, kb, b, (int)threadIdx.x);
    }
}
__global__ void __launch_bounds__(256) qmm_epilogue_kernel(const QmmArgs a, const float* __restrict__ part, const int ldp,
                                                           const int ks, const int BP, const float* __restrict__ ssp,
                                                           const QmgChainOut ch, const float* __restrict__ rscale = nullptr) {
    qmm_kernarg_burst_all<true>(a);
    if (a.rows_dev && *a.rows_dev <= a.rows_min) return;
    qmm_epilogue_body(a, part, ldp, ks, BP, ssp, ch, rscale, 0, a.B);
}
// grouped launches (QwGroup, qmm_wide1.inc): blockIdx.z = the group (STORE / SILU_MUL epilogues: the experts of a layer)
__global__ void __launch_bounds__(256) qmm_epilogue_grp_kernel(const QmmArgs a, const float* __restrict__ part, const int ldp,
                                                               const int ks, const int BP, const float* __restrict__ ssp,
                                                               const QmgChainOut ch0, const QwGroup g) {
    qmm_kernarg_burst_all<true>(a);
    const int e = blockIdx.z;
    const int nb = a.rows_dev ? min(a.B, a.rows_dev[e] - a.rows_min) : a.B;
    // rows past the group's count: nothing to sum, and no image entry either -- every kernel of the path keeps rows apart (a row of the
    // next mat-mul's image only ever meets its own scale, sums and accumulators), so what the next launch finds there is never stored
    if ((int)blockIdx.y >= nb) return;
    QmgChainOut ch = ch0;
    if (ch.img) {
        ch.img += (size_t)e * g.chimg;
        ch.ssp = reinterpret_cast<float*>(reinterpret_cast<uint8_t*>(ch.ssp) + (size_t)e * g.chimg);
    }
    qmm_epilogue_body(a, part + (size_t)e * g.part, ldp, ks, BP,
                      reinterpret_cast<const float*>(reinterpret_cast<const uint8_t*>(ssp) + (size_t)e * g.img), ch, nullptr, (size_t)e * g.out, nb);
}

// (Round 4 measured a form of this kernel with FOUR token rows per workgroup -- a quarter of the workgroups, all partial sums requested up
//  front -- on the theory that 512-3584 tiny workgroups per launch are bound by the rate at which workgroups start: it LOST, 10.4 / 13.0 /
//  9.1 us against 5.7 / 7.5 / 8.0 us for the wo | down, q|k|v and gate/up epilogues, the ragged batch-32 step 5690 against 6255 tok/s
//  (profiles/r04_b32_epilogue_rows_ab.txt).  The epilogue is a chain of three dependent memory round trips; more rows per thread make
//  the chain longer, more workgroups hide it.  Deleted.)
#include "qmm_wide1_gemm.inc"

// Scratch of the wide / prompt paths (activation images, split-K partials, the prompt-step workspace) belongs to the
// (device, stream) that launches: two models on two streams never share an image, growing a buffer never frees memory a
// captured graph still replays from (scratch.cpp).  The chain hint is per (device, stream) too.
// image staged by the last chained epilogue: valid for exactly the next wide launch if it consumes the same activations
struct QmgChainState { bool valid; const void* x; int B, K, MT, buf; const float* norm_w; hipStream_t st; int sp; int x_dtype = MI355_DTYPE_F32;
                       int grp = 1; int64_t grp_stride = 0;                       // grouped launch: groups and elements between their activations
                       // the images were built WITHOUT their activations ever being written to `x` (mi355_internal_moe_stage_grouped: `x` is
                       // only the key the consumer will present).  A launch that presents this key and cannot take the images must fail, it
                       // must never stage from `x` (stale rows, silently wrong results -- ADVICE r5)
                       bool prestaged = false; };
struct QmgStream { int cur = 0; QmgChainState chain = {false, nullptr, 0, 0, 0, 0, nullptr, nullptr, 0}; };
static std::mutex g_qmg_mu;
static std::map<std::pair<int, hipStream_t>, QmgStream> g_qmg_streams;
static QmgStream& qmg_stream(hipStream_t st) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(g_qmg_mu);
    return g_qmg_streams[std::make_pair(dev, st)];              // std::map: the reference stays valid across later inserts
}
// ONE statement of "a grouped mat-mul of B token rows runs as the z extent of the 9..32-token launches" (mi355_qmm_launch takes that path,
// mi355_internal_moe_stage_grouped builds images only for it, mi355_internal_moe_scatter_combine_to_image stages only for it): three hand
// copies of this condition had to agree for the fused MoE staging to be correct.  Defined below the tuning switches.
static bool qmg_wide_one_launch(int B);
static int g_tune_qmv = 0;                                    // mi355_set_tuning(20, 1): single-token launches take the LDS-DMA engine (qmv_engine.inc) one by one -- measured slower than qmm_kernel per launch (fixed cost), faster chained
static int g_tune_qmv_nc = 8;                                 // mi355_set_tuning(21, n): consumer waves per workgroup of the engine (1..15)
static int g_tune_exact_act = 0;                              // mi355_set_tuning(24, 1): "exact" activations on the 9..32-token and prompt paths: f16 hi + lo planes (22 bits) instead of one f16 plane
static int g_tune_chain_b1 = 0;                               // mi355_set_tuning(23, 1): probe builds: chain the single-token mat-vecs between two attention calls into one persistent launch
static int g_tune_qmv_ring = 0;                               // mi355_set_tuning(22, 64): the engine never takes the 128 KiB ring (A/B)
static int g_tune_chain = 1;                                  // mi355_set_tuning(9, 0): never chain (A/B experiments)
static int g_tune_wide16 = 0;                                 // mi355_set_tuning(15, n): launches of >= n (row tile x k-block) units use 16-wave workgroups
static int g_tune_merge = 1;                                  // mi355_set_tuning(14, 0): one launch per run of same-type segments (A/B)
static int g_tune_ks_minkb = 2;                               // mi355_set_tuning(17, n): fewest k-blocks a k-split keeps per workgroup
static int g_tune_grp_ks_target = 4096;                       // mi355_set_tuning(16, n): the same for grouped launches (slots of all groups)
static int g_tune_ks_target = 1024;                           // mi355_set_tuning(10, n): split K until a launch has n row-tile x k-split slots
static int g_tune_wide_fuse = 0;                              // mi355_set_tuning(49, 1), probe builds only: EXPERIMENT, the 9..32-token path runs its split-K epilogue inside the GEMM launches (lost its A/B: qmm_wide1_gemm.inc)
static inline int qmg_buf(void** p, int key, size_t need, hipStream_t st) { return mi355_scratch_get(p, key, need, st, false); }

template <int MT>
static int qmg_launch(const QmmArgs& a0, hipStream_t st) {
    QmmArgs a = a0;
    if (a.paired) a.paired = 0;                 // slots are plain (segment, tile) order; the epilogue pairs gate/up rows
    const int nkb = a.K / 256;
    int n_slots = 0;
    for (int s = 0; s < a.nseg; ++s) n_slots += a.seg[s].n_tiles;
    const int ldp = n_slots * 16;
    const size_t kbb = qmg_kb_bytes(MT);
    // split K until the launch has >= 1024 (row tile, k-split) slots = 4 consumer waves per CU, keeping >= 2 k-blocks per
    // workgroup.  (Measured at batch 32: targets 512 / 768 / 1024 / 1280 / 1536 / 2048 / 4096 -> 4081 / 4634 / 4835-4900 /
    // 4737 / 4742 / 4686 / 4607 tok/s: less splitting = fewer partial sums for the epilogue and longer pipelines.)
    const bool wide16 = g_tune_wide16 > 0 && (int64_t)n_slots * nkb >= g_tune_wide16;
    const int NCq = wide16 ? QMG_NC_WIDE : QMG_NC;
    int ks = 1;
    if (wide16) { while ((n_slots + NCq - 1) / NCq * ks < 200 && nkb / (ks * 2) >= 4) ks *= 2; }
    else if (g_tune_ks_target > 0) { while (n_slots * ks < g_tune_ks_target && nkb / (ks * 2) >= g_tune_ks_minkb) ks *= 2; }
    else { while ((n_slots + QMG_NC - 1) / QMG_NC * ks < -g_tune_ks_target && nkb / (ks * 2) >= g_tune_ks_minkb) ks *= 2; }
    // activation image: staged by the previous launch's epilogue (chain) or by the prep kernel now
    QmgStream& qs = qmg_stream(st);
    const bool chained = qs.chain.valid && qs.chain.x == a.x && qs.chain.B == a.B && qs.chain.K == a.K &&
                         qs.chain.MT == MT && qs.chain.sp == 0 && qs.chain.norm_w == a.norm_w && qs.chain.st == st &&
                         a.x_dtype == MI355_DTYPE_F32 && a.ldx == a.K;
    qs.chain.valid = false;
    const int cur = chained ? qs.chain.buf : qs.cur;
    int rc = 0;
    void* imgp = nullptr;
    // (a chained image was sized by the epilogue that staged it: the same request returns the same buffer)
    rc = qmg_buf(&imgp, cur ? MI355_SCR_QMM_IMG1 : MI355_SCR_QMM_IMG0, kbb * nkb + (size_t)nkb * MT * 8 * sizeof(float), st);
    if (rc) return rc;
    void* partp = nullptr;
    rc = qmg_buf(&partp, MI355_SCR_QMM_PART, (size_t)ks * MT * 8 * ldp * sizeof(float), st);
    if (rc) return rc;
    float* const part = static_cast<float*>(partp);
    static Mi355DevOnce attr_done;
    if (!attr_done.done()) {
        (void)hipFuncSetAttribute((const void*)qmm_gemm_kernel<MT, MI355_GGML_Q4_K>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)qmm_gemm_kernel<MT, MI355_GGML_Q6_K>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)qmm_gemm2_kernel<MT, MI355_GGML_Q4_K, MI355_GGML_Q6_K>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)qmm_gemm16_kernel<MT, MI355_GGML_Q4_K>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)qmm_gemm16_kernel<MT, MI355_GGML_Q6_K>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done.set();
    }
    uint8_t* img = static_cast<uint8_t*>(imgp);
    float* ssp = reinterpret_cast<float*>(img + kbb * nkb);                 // [nkb][MT*8] after the image
    if (!chained) hipLaunchKernelGGL((qmm_prep2_kernel<MT>), dim3(nkb, MT), dim3(256), 0, st, img, ssp, a, kbb);
    // one GEMM launch per run of same-type segments: the type-specialised builds need no schedule pinning and do not
    // spill (the mixed build did); each run writes its own columns of the partial-sum buffer
    int s_split = 0, slots0 = 0;                                     // exactly one Q4_K run followed by one Q6_K run: one launch
    while (s_split < a.nseg && a.seg[s_split].type == MI355_GGML_Q4_K) slots0 += a.seg[s_split++].n_tiles;
    bool two_runs = g_tune_merge && !wide16 && s_split > 0 && s_split < a.nseg;
    for (int q = s_split; q < a.nseg; ++q) two_runs = two_runs && a.seg[q].type == MI355_GGML_Q6_K;
    if (two_runs) {
        const int slots1 = n_slots - slots0;
        const dim3 ggrid((slots0 + QMG_NC - 1) / QMG_NC + (slots1 + QMG_NC - 1) / QMG_NC, ks);
        hipLaunchKernelGGL((qmm_gemm2_kernel<MT, MI355_GGML_Q4_K, MI355_GGML_Q6_K>), ggrid, dim3(512), 2 * kbb, st, a, img, part, ldp, s_split, slots0, slots1);
    }
    for (int s0 = 0, slot_base = 0; s0 < a.nseg && !two_runs;) {
        int s1 = s0 + 1;
        while (s1 < a.nseg && a.seg[s1].type == a.seg[s0].type) ++s1;
        QmmArgs r = a;
        r.nseg = s1 - s0;
        int run_slots = 0;
        for (int q = 0; q < r.nseg; ++q) { r.seg[q] = a.seg[s0 + q]; run_slots += r.seg[q].n_tiles; }
        const dim3 ggrid((run_slots + NCq - 1) / NCq, ks);
        if (wide16 && r.seg[0].type == MI355_GGML_Q4_K)
            hipLaunchKernelGGL((qmm_gemm16_kernel<MT, MI355_GGML_Q4_K>), ggrid, dim3(1024), 2 * kbb, st, r, img, part, ldp, run_slots, slot_base);
        else if (wide16)
            hipLaunchKernelGGL((qmm_gemm16_kernel<MT, MI355_GGML_Q6_K>), ggrid, dim3(1024), 2 * kbb, st, r, img, part, ldp, run_slots, slot_base);
        else if (r.seg[0].type == MI355_GGML_Q4_K)
            hipLaunchKernelGGL((qmm_gemm_kernel<MT, MI355_GGML_Q4_K>), ggrid, dim3(512), 2 * kbb, st, r, img, part, ldp, run_slots, slot_base);
        else
            hipLaunchKernelGGL((qmm_gemm_kernel<MT, MI355_GGML_Q6_K>), ggrid, dim3(512), 2 * kbb, st, r, img, part, ldp, run_slots, slot_base);
        slot_base += run_slots;
        s0 = s1;
    }
    // chain: this epilogue also stages the image of the next wide mat-mul (x = our out) into the other buffer
    QmgChainOut ch{nullptr, nullptr, nullptr, 0, 0, 0, 0};
    const bool want = g_tune_chain && a.chain_next && a.next_k > 0 && (a.next_k % 256) == 0 && a.ldo == a.next_k &&
                      (a.epi == MI355_EPI_RESID || a.epi == MI355_EPI_SILU_MUL || a.epi == MI355_EPI_STORE) &&
                      (a.epi == MI355_EPI_SILU_MUL ? a.seg[0].n_rows == a.next_k : (a.nseg == 1 && a.seg[0].n_rows == a.next_k));
    if (want) {
        const int other = cur ^ 1, nkb2 = a.next_k / 256;
        void* onext = nullptr;
        rc = qmg_buf(&onext, other ? MI355_SCR_QMM_IMG1 : MI355_SCR_QMM_IMG0, kbb * nkb2 + (size_t)nkb2 * MT * 8 * sizeof(float), st);
        if (rc) return rc;
        uint8_t* oimg = static_cast<uint8_t*>(onext);
        ch = QmgChainOut{oimg, reinterpret_cast<float*>(oimg + kbb * nkb2), a.next_norm_w, a.next_k, MT, kbb, 0};
        qs.chain = QmgChainState{true, a.out, a.B, a.next_k, MT, other, a.next_norm_w, st, 0};
    }
    qs.cur = cur;
    hipLaunchKernelGGL(qmm_epilogue_kernel, dim3((ldp + 255) / 256, want ? MT * 8 : a.B), dim3(256), 0, st, a, part, ldp, ks,
                       MT * 8, ssp, ch);
    return (int)hipGetLastError();
}

// single-plane launcher (qmm_wide1.inc, qmm_wide1_gemm.inc): MT = m-tiles of 16 tokens (1: 9..16 tokens, 2: 17..32)
template <int MT>
static int qw1_launch(const QmmArgs& a0, hipStream_t st) {
    QmmArgs a = a0;
    if (a.paired) a.paired = 0;
    const int nkb = a.K / 256;
    int n_slots = 0;
    for (int s = 0; s < a.nseg; ++s) n_slots += a.seg[s].n_tiles;
    const int ldp = n_slots * 16, BP = MT * 16;
    const size_t kbb = qw1_kb_bytes(MT);
    // grouped launch: G mat-muls of these shapes in the z extent of every launch (the experts of a mixture-of-experts layer)
    const int G = a.grp_n > 1 ? a.grp_n : 1;
    int ks = 1;
    if (G > 1) { while ((int64_t)n_slots * G * ks < g_tune_grp_ks_target && nkb / (ks * 2) >= g_tune_ks_minkb) ks *= 2; }
    else if (g_tune_ks_target > 0) { while (n_slots * ks < g_tune_ks_target && nkb / (ks * 2) >= g_tune_ks_minkb) ks *= 2; }
    else {
        // workgroup target: the smallest split (any integer, not only powers of two) that puts >= |target| workgroups on the chip while
        // every split keeps >= minkb k-blocks; e.g. the 256-tile down projection: 37 workgroups x 7 splits of 8 k-blocks = 259
        const int n_wg = (n_slots + QMG_NC - 1) / QMG_NC;
        int want = (-g_tune_ks_target + n_wg - 1) / n_wg;
        const int ks_max = nkb / g_tune_ks_minkb > 1 ? nkb / g_tune_ks_minkb : 1;
        if (want > ks_max) want = ks_max;
        if (want < 1) want = 1;
        ks = want;
    }
#ifndef QW1_Q6K_LONGK_WGS
#define QW1_Q6K_LONGK_WGS 0
#endif
    if (QW1_Q6K_LONGK_WGS > 0 && G == 1 && g_tune_ks_target > 0 && a.nseg == 1 && nkb >= 32 && a.seg[0].type == MI355_GGML_Q6_K) {
        // Rounds 4-5 (two MFMA chains per Q6_K weight, 411 VALU per tile and k-block: profiles/r03_pmc_sq_b32_set1.json): the long-K Q6_K
        // launch (the down projection: 256 tiles x 56 k-blocks) was bound by its unpack arithmetic, and 37 x 14 = 518 workgroups -- two per
        // CU, the second one's waves in the first's stalls -- beat the power-of-two split (37 x 4): 27.6 -> 23.8 us.  With ONE operand per
        // weight (round 6, QW1_Q6K_ONE_PLANE) the arithmetic is no longer the bound and the 14 partial sums per output are pure cost:
        // same box, ragged batch-32 step: 14 splits 6515 / 6529 tok/s, 7 splits 6699 / 6667, the general rule's 4 splits 6749 / 6729
        // (profiles/r06_b32_q6k_ab.txt) -- the special case is off (-DQW1_Q6K_LONGK_WGS=512 builds it back for A/B).
        const int n_wg = (n_slots + QMG_NC - 1) / QMG_NC;
        const int cus = QW1_Q6K_LONGK_WGS;
        if (n_wg * ks < cus) {
            int want = (cus + n_wg - 1) / n_wg;
            while (want > 1 && nkb / want < 4) --want;
            if (want > ks) ks = want;
        }
    }
    // (gate/up, 256 workgroups of 16 k-blocks: a split into 2 x 8 for two workgroups per CU measured +-0.1 % on the batch-32 step, round 4)
    {   // no empty split: the kernels walk ceil(nkb / ks) k-blocks per split, and every split's partial sums are added up
        const int kb_per = (nkb + ks - 1) / ks;
        ks = (nkb + kb_per - 1) / kb_per;
    }
    QmgStream& qs = qmg_stream(st);
    const bool chained = qs.chain.valid && qs.chain.x == a.x && qs.chain.B == a.B && qs.chain.K == a.K &&
                         qs.chain.MT == MT && qs.chain.sp == 1 && qs.chain.norm_w == a.norm_w && qs.chain.st == st &&
                         a.x_dtype == qs.chain.x_dtype && a.ldx == a.K && qs.chain.grp == G &&
                         (G == 1 || qs.chain.grp_stride == a.grp_x);
    const bool orphan = qs.chain.valid && qs.chain.prestaged && qs.chain.x == a.x && !chained;
    qs.chain.valid = false;
    if (orphan) {                                                       // the images behind this key were staged for another launch shape,
        mi355_note_error((int)hipErrorInvalidValue);                     // and `x` itself was never written: refuse instead of staging stale rows
        return (int)hipErrorInvalidValue;
    }
    const int cur = chained ? qs.chain.buf : qs.cur;
    int rc = 0;
    void* imgp = nullptr;
    // (a group's image + sum-of-squares partials, rounded to the DMA granule: group e's lie e * imgb bytes after group 0's)
    const size_t imgb = (kbb * nkb + (size_t)nkb * BP * sizeof(float) + 1023) / 1024 * 1024;
    rc = qmg_buf(&imgp, cur ? MI355_SCR_QMM_IMG1 : MI355_SCR_QMM_IMG0, G > 1 ? G * imgb : kbb * nkb + (size_t)nkb * BP * sizeof(float), st);
    if (rc) return rc;
    void* partp = nullptr;
    rc = qmg_buf(&partp, MI355_SCR_QMM_PART, (size_t)G * ks * BP * ldp * sizeof(float), st);
    if (rc) return rc;
    float* const part = static_cast<float*>(partp);
    static Mi355DevOnce attr_done;
    if (!attr_done.done()) {
        (void)hipFuncSetAttribute((const void*)qw1_gemm_kernel<MT, MI355_GGML_Q4_K>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)qw1_gemm_kernel<MT, MI355_GGML_Q6_K>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)qw1_gemm2_kernel<MT, MI355_GGML_Q4_K, MI355_GGML_Q6_K>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)qw1_gemm_grp_kernel<MT, MI355_GGML_Q4_K>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)qw1_gemm_grp_kernel<MT, MI355_GGML_Q6_K>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)qw1_gemm2_grp_kernel<MT, MI355_GGML_Q4_K, MI355_GGML_Q6_K>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done.set();
    }
    uint8_t* img = static_cast<uint8_t*>(imgp);
    float* ssp = reinterpret_cast<float*>(img + kbb * nkb);                 // [nkb][BP] after the image
    // chain: the epilogue also stages the image of the next wide mat-mul (x = our out) into the other buffer
    QmgChainOut ch{nullptr, nullptr, nullptr, 0, 0, 0, 0};
    size_t imgb2 = 0;
    const bool want = g_tune_chain && a.chain_next && a.next_k > 0 && (a.next_k % 256) == 0 && a.ldo == a.next_k &&
                      (a.epi == MI355_EPI_RESID || a.epi == MI355_EPI_SILU_MUL || a.epi == MI355_EPI_STORE) &&
                      (a.epi == MI355_EPI_SILU_MUL ? a.seg[0].n_rows == a.next_k : (a.nseg == 1 && a.seg[0].n_rows == a.next_k));
    if (want) {
        const int other = cur ^ 1, nkb2 = a.next_k / 256;
        void* onext = nullptr;
        imgb2 = (kbb * nkb2 + (size_t)nkb2 * BP * sizeof(float) + 1023) / 1024 * 1024;
        rc = qmg_buf(&onext, other ? MI355_SCR_QMM_IMG1 : MI355_SCR_QMM_IMG0, G > 1 ? G * imgb2 : kbb * nkb2 + (size_t)nkb2 * BP * sizeof(float), st);
        if (rc) return rc;
        uint8_t* oimg = static_cast<uint8_t*>(onext);
        ch = QmgChainOut{oimg, reinterpret_cast<float*>(oimg + kbb * nkb2), a.next_norm_w, a.next_k, MT, kbb, 1};
    }
    const QwGroup grp{a.grp_x, a.grp_out, (int64_t)imgb, (int64_t)ks * BP * ldp, (int64_t)imgb2};
    // the split-K epilogue inside the GEMM launches (qmm_wide1_gemm.inc): tickets per 256-column block of the output
    QwFuse fz{};
    fz.ticket = nullptr; fz.ssp = ssp; fz.ch = ch; fz.ks = ks; fz.n_runs = 0;
    if (QW1_FUSE_BUILD && g_tune_wide_fuse && G == 1) {
        void* tk = nullptr;
        rc = mi355_scratch_get(&tk, MI355_SCR_QMM_TICKET, 8192 * sizeof(unsigned), st, true);
        if (rc) return rc;
        if ((ldp + 255) / 256 <= 8192) fz.ticket = static_cast<unsigned*>(tk);
    }
    if (!chained && G > 1) hipLaunchKernelGGL((qw1_prep_grp_kernel<MT>), dim3(nkb, MT * 2, G), dim3(256), 0, st, img, ssp, a, kbb, grp);
    else if (!chained) hipLaunchKernelGGL((qw1_prep_kernel<MT>), dim3(nkb, MT * 2), dim3(256), 0, st, img, ssp, a, kbb);
    // per-tile epilogue inside the GEMM launches (qmm_wide1_gemm.inc, TEPI): mat-muls that stage no image for a successor -- q|k|v (RoPE +
    // cache write), the lm_head, unchained stores / residual adds.
    // EXPERIMENT, probe builds only (key 49 = 2): measured in round 4 and lost -- q|k|v 40.7 us against 12.7 + 7.5 us for GEMM + epilogue
    // launch, the lm_head 158 against 125 + 10 us, the ragged batch-32 step 5336 against 6130 tok/s (profiles/r04_b32_tile_epilogue_ab.txt):
    // a wave that applies RoPE + the cache write to its eight elements walks eight chains of dependent loads (position -> cos / sin ->
    // slot -> store) one after the other, where the epilogue launch gives every element its own thread.
    bool tepi = false;
#if QW1_FUSE_BUILD
    tepi = G == 1 && g_tune_wide_fuse == 2 && !fz.ticket && !want && n_slots <= 8192 &&
           (a.epi == MI355_EPI_QKV_ROPE_CACHE || a.epi == MI355_EPI_STORE || a.epi == MI355_EPI_RESID);
    if (tepi) {
        void* tk = nullptr;
        rc = mi355_scratch_get(&tk, MI355_SCR_QMM_TICKET, 8192 * sizeof(unsigned), st, true);
        if (rc) return rc;
        fz.tile_ticket = static_cast<unsigned*>(tk);
        static bool tepi_attr = false;
        if (!tepi_attr) {
            (void)hipFuncSetAttribute((const void*)qw1_gemm_tepi_kernel<MT, MI355_GGML_Q4_K>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)qw1_gemm_tepi_kernel<MT, MI355_GGML_Q6_K>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)qw1_gemm2_tepi_kernel<MT, MI355_GGML_Q4_K, MI355_GGML_Q6_K>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            tepi_attr = true;
        }
    }
#endif
    int s_split = 0, slots0 = 0;                                     // exactly one Q4_K run followed by one Q6_K run: one launch
    while (s_split < a.nseg && a.seg[s_split].type == MI355_GGML_Q4_K) slots0 += a.seg[s_split++].n_tiles;
    bool two_runs = g_tune_merge && s_split > 0 && s_split < a.nseg;
    for (int q = s_split; q < a.nseg; ++q) two_runs = two_runs && a.seg[q].type == MI355_GGML_Q6_K;
    // the runs of same-type segments = the launches (or launch halves): every workgroup needs all of them to count a block's arrivals
    for (int s0 = 0, base = 0; s0 < a.nseg;) {
        int s1 = s0 + 1, rs = a.seg[s0].n_tiles;
        while (s1 < a.nseg && a.seg[s1].type == a.seg[s0].type) rs += a.seg[s1++].n_tiles;
        fz.run_lo[fz.n_runs] = base; fz.run_hi[fz.n_runs] = base + rs; ++fz.n_runs;
        base += rs;
        s0 = s1;
    }
    if (two_runs) {
        const int slots1 = n_slots - slots0;
        const dim3 ggrid((slots0 + QMG_NC - 1) / QMG_NC + (slots1 + QMG_NC - 1) / QMG_NC, ks, G);
        if (G > 1) hipLaunchKernelGGL((qw1_gemm2_grp_kernel<MT, MI355_GGML_Q4_K, MI355_GGML_Q6_K>), ggrid, dim3(512), QW1_NB * kbb, st, a, img, part, ldp, s_split, slots0, slots1, fz, grp);
        else
#if QW1_FUSE_BUILD
        if (tepi) hipLaunchKernelGGL((qw1_gemm2_tepi_kernel<MT, MI355_GGML_Q4_K, MI355_GGML_Q6_K>), ggrid, dim3(512), QW1_NB * kbb + 256, st, a, img, part, ldp, s_split, slots0, slots1, fz);
        else
#endif
        hipLaunchKernelGGL((qw1_gemm2_kernel<MT, MI355_GGML_Q4_K, MI355_GGML_Q6_K>), ggrid, dim3(512), QW1_NB * kbb, st, a, img, part, ldp, s_split, slots0, slots1, fz);
    }
    for (int s0 = 0, slot_base = 0; s0 < a.nseg && !two_runs;) {
        int s1 = s0 + 1;
        while (s1 < a.nseg && a.seg[s1].type == a.seg[s0].type) ++s1;
        QmmArgs r = a;
        r.nseg = s1 - s0;
        int run_slots = 0;
        for (int q = 0; q < r.nseg; ++q) { r.seg[q] = a.seg[s0 + q]; r.moe_stride[q] = a.moe_stride[s0 + q]; run_slots += r.seg[q].n_tiles; }
        // (the in-launch epilogue walks ALL segments of the mat-mul: it gets the whole descriptor, the GEMM part its own run)
        const dim3 ggrid((run_slots + QMG_NC - 1) / QMG_NC, ks, G);
        if (G > 1) {
            if (r.seg[0].type == MI355_GGML_Q4_K)
                hipLaunchKernelGGL((qw1_gemm_grp_kernel<MT, MI355_GGML_Q4_K>), ggrid, dim3(512), QW1_NB * kbb, st, r, img, part, ldp, run_slots, slot_base, fz, grp);
            else
                hipLaunchKernelGGL((qw1_gemm_grp_kernel<MT, MI355_GGML_Q6_K>), ggrid, dim3(512), QW1_NB * kbb, st, r, img, part, ldp, run_slots, slot_base, fz, grp);
        } else
#if QW1_FUSE_BUILD
        if (fz.ticket && r.nseg != a.nseg) {
            // a mat-mul of several launches: the kernel's segment walk starts at the run's first segment
            QmmArgs full = a;
            if (r.seg[0].type == MI355_GGML_Q4_K)
                hipLaunchKernelGGL((qw1_gemm_run_kernel<MT, MI355_GGML_Q4_K>), ggrid, dim3(512), QW1_NB * kbb, st, full, img, part, ldp, run_slots, slot_base, s0, s1, fz);
            else
                hipLaunchKernelGGL((qw1_gemm_run_kernel<MT, MI355_GGML_Q6_K>), ggrid, dim3(512), QW1_NB * kbb, st, full, img, part, ldp, run_slots, slot_base, s0, s1, fz);
        } else
#endif
#if QW1_FUSE_BUILD
        if (tepi) {                                                   // the whole descriptor: the epilogue needs every segment's rows
            if (r.seg[0].type == MI355_GGML_Q4_K)
                hipLaunchKernelGGL((qw1_gemm_tepi_kernel<MT, MI355_GGML_Q4_K>), ggrid, dim3(512), QW1_NB * kbb + 256, st, a, img, part, ldp, run_slots, slot_base, s0, s1, fz);
            else
                hipLaunchKernelGGL((qw1_gemm_tepi_kernel<MT, MI355_GGML_Q6_K>), ggrid, dim3(512), QW1_NB * kbb + 256, st, a, img, part, ldp, run_slots, slot_base, s0, s1, fz);
        } else
#endif
        if (r.seg[0].type == MI355_GGML_Q4_K)
            hipLaunchKernelGGL((qw1_gemm_kernel<MT, MI355_GGML_Q4_K>), ggrid, dim3(512), QW1_NB * kbb, st, r, img, part, ldp, run_slots, slot_base, fz);
        else
            hipLaunchKernelGGL((qw1_gemm_kernel<MT, MI355_GGML_Q6_K>), ggrid, dim3(512), QW1_NB * kbb, st, r, img, part, ldp, run_slots, slot_base, fz);
        slot_base += run_slots;
        s0 = s1;
    }
    if (want) { qs.chain = QmgChainState{true, a.out, a.B, a.next_k, MT, cur ^ 1, a.next_norm_w, st, 1}; qs.chain.grp = G; qs.chain.grp_stride = a.grp_out; }
    qs.cur = cur;
    if (G > 1)
        hipLaunchKernelGGL(qmm_epilogue_grp_kernel, dim3((ldp + 255) / 256, want ? BP : a.B, G), dim3(256), 0, st, a, part, ldp, ks, BP, ssp, ch, grp);
    else if (!fz.ticket && !tepi)
        hipLaunchKernelGGL(qmm_epilogue_kernel, dim3((ldp + 255) / 256, want ? BP : a.B), dim3(256), 0, st, a, part, ldp, ks, BP, ssp, ch);
    return (int)hipGetLastError();
}

// ---- the merge of the balanced attention stream's partials, fused with the staging of the NEXT mat-mul's activation image (round 4).
// At batch 9..32 the attention output feeds wo through the 9..32-token path, whose image (one f16 plane, block scale per token and
// k-block) was built by a launch of its own (qw1_prep_kernel, 4.9 us per layer) right after the merge launch (4.7 us): a k-block = 256
// columns = two heads of 128 channels, so a workgroup that merges two heads of one sequence holds exactly one image row.  Same merge
// arithmetic as paged_attn_stream_reduce_kernel, same entry builder as the staging kernel on the bf16-rounded outputs: bit-identical
// image, one launch and one boundary fewer per layer.
#include "pa_stream_cut.h"
template <int MT>
__global__ void __launch_bounds__(256) pa_stream_reduce_img_kernel(uint16_t* __restrict__ out, const float* __restrict__ tmp_out,
                                                                   const float* __restrict__ max_logits, const float* __restrict__ exp_sums,
                                                                   const uint32_t* __restrict__ context_lens, const int B, const int H,
                                                                   const int W, const int slots, uint8_t* __restrict__ img,
                                                                   float* __restrict__ ssp, const size_t kbb) {
    constexpr int D = 128;
    const int kb = blockIdx.x, b = blockIdx.y;
    const int lane = threadIdx.x & 63, h = 2 * kb + (threadIdx.x >> 7), d = threadIdx.x & 127;
    __shared__ float sm_o[256];
    float v = 0.f;
    if (b < B) {
        const int n_l = lane < B ? ((int)context_lens[lane] + 63) >> 6 : 0;
        int pend = n_l;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(pend, o, 64);
            if (lane >= o) pend += t;
        }
        const int S = __builtin_amdgcn_readlane(pend, 63);
        const int pe = __builtin_amdgcn_readlane(pend, b), ps = pe - __builtin_amdgcn_readlane(n_l, b);
        bool meets = false;
        if (lane < W) {
            const int lo = (int)pas_cut(S, W, lane), hi = (int)pas_cut(S, W, lane + 1);
            meets = lo < hi && lo < pe && hi > ps;
        }
        const uint64_t mask = __ballot(meets);
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
        uint16_t ob = 0;
        if (mask) {
            ob = f32_to_bf16(den > 0.f ? acc / den : 0.f);
            out[((int64_t)b * H + h) * D + d] = ob;
        } else {
            ob = out[((int64_t)b * H + h) * D + d];                     // a sequence without stages: the output row is left as it is
        }
        v = bf16_to_f32(ob);                                          // wo reads the bf16 tensor (attention.rs:997-1004)
    }
    sm_o[threadIdx.x] = v;
    __syncthreads();
    if (threadIdx.x < 32) {
        float e8[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) e8[i] = sm_o[threadIdx.x * 8 + i];
        qw1_prep_entry(img, ssp, e8, b < B, nullptr, MT, kbb, kb, b, (int)threadIdx.x);
    }
}
/* host layers: merge the stream's partials (mi355_internal_paged_attention_v2_partials) into `out` bf16 [B, H, 128] AND stage the image of
 * the wide mat-mul that reads `out` next on this stream (x = out, k = H * 128, no norm).  9 <= B <= 32, H even.  -4: not this shape. */
extern "C" int mi355_internal_pa_stream_reduce_to_image(void* out, const float* tmp_out, const float* max_logits, const float* exp_sums,
                                                        const uint32_t* context_lens, int32_t B, int32_t H, int32_t W, int32_t slots,
                                                        int64_t stream) {
    if (B < 9 || B > 32 || (H & 1) || W < 1 || W > 64 || g_tune_exact_act || !g_tune_chain) return -4;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    const int MT = B <= 16 ? 1 : 2, BP = MT * 16, K = H * 128, nkb = K / 256;
    const size_t kbb = qw1_kb_bytes(MT);
    QmgStream& qs = qmg_stream(st);
    const int other = qs.cur ^ 1;
    void* p = nullptr;
    const int rc = qmg_buf(&p, other ? MI355_SCR_QMM_IMG1 : MI355_SCR_QMM_IMG0, kbb * nkb + (size_t)nkb * BP * sizeof(float), st);
    if (rc) return rc;
    uint8_t* img = static_cast<uint8_t*>(p);
    float* ssp = reinterpret_cast<float*>(img + kbb * nkb);
    const dim3 grid(H / 2, BP);
    if (MT == 1)
        hipLaunchKernelGGL((pa_stream_reduce_img_kernel<1>), grid, dim3(256), 0, st, static_cast<uint16_t*>(out), tmp_out, max_logits, exp_sums,
                           context_lens, B, H, W, slots, img, ssp, kbb);
    else
        hipLaunchKernelGGL((pa_stream_reduce_img_kernel<2>), grid, dim3(256), 0, st, static_cast<uint16_t*>(out), tmp_out, max_logits, exp_sums,
                           context_lens, B, H, W, slots, img, ssp, kbb);
    QmgChainState cs{true, out, B, K, MT, other, nullptr, st, 1};
    cs.x_dtype = MI355_DTYPE_BF16;
    qs.chain = cs;
    return (int)hipGetLastError();
}

// ================================================================================================
// Prompt-step path (num_tokens >= QMP_MIN_TOKENS): the matmul is compute-bound, so it is run as a plain GEMM on the
// matrix cores instead of streaming the quantised weights once per 32 tokens.
//   1. `qmp_dequant_kernel`  : repacked Q4_K / Q6_K tiles -> bf16 hi + bf16 lo, row-major [N, K] (w = hi + lo to 2^-17)
//   2. `qmp_xsplit_kernel`   : activations (RMSNorm applied here) -> bf16 hi + lo, row-major [T, K]
//   3. three library GEMMs (rocBLAS, bf16 inputs, f32 accumulate): hi.hi + hi.lo + lo.hi  -> f32 [T, N]
//      (candle's own GPU path dequantises to f16 and calls the vendor GEMM for prompts [EXT]; the hi/lo split keeps
//       the result at f32-activation accuracy so the same 1e-3 logit bound holds as for decode)
//   4. `qmm_epilogue_kernel` : the same fused epilogues (bias / residual / SiLU*mul / RoPE + cache scatter)
// rocBLAS is bound with dlopen (the copy torch already loaded if there is one), like RCCL in host_model.cpp.
#include <dlfcn.h>
#define QMP_MIN_TOKENS 96

#ifdef MI355_QMM_PROBES   // probe builds only (tools/build_probe_lib.sh): first-generation prompt path: weight image for library GEMMs
#include "probes/qmp_kernels.inc"
#endif

#ifdef MI355_QMM_PROBES   // probe builds only (tools/build_probe_lib.sh): rocBLAS by dlopen (A/B partner of the hand-written prompt GEMMs)
#include "probes/qmp_rocblas.inc"
#endif

static int g_tune_qpg = 2;                                 // mi355_set_tuning(11, v): prompt-step GEMM workgroup tile: 2 (default since round 4) = 64 tokens x 256 rows, 0 = 32 x 512 (rounds 2-3), 1 / 3: A/B variants
static int g_tune_qpg_fepi = 1;                            // mi355_set_tuning(48, 0): A/B, separate epilogue launch.  Default: the prompt-step GEMM applies the epilogue itself (Q4_K launches; store / residual / SiLU * up; round 4: +7 % on the prompt step)
static int g_tune_qpg_min = 96;                            // mi355_set_tuning(12, n): fewest tokens that take the prompt-step GEMM (QMP_MIN_TOKENS)
static int g_tune_dbg = 0;                                 // mi355_set_tuning(2, v): probe / ablation modes (experiments only)

#include "qmm_prefill.inc"

// hand-written prompt-step GEMM (qmm_prefill.inc); returns hipErrorNotSupported for launches it does not cover (then the
// caller keeps the library-GEMM path)
static int qpg_launch(const QmmArgs& a0, hipStream_t st) {
    QmmArgs a = a0;
    a.paired = 0;
    for (int s = 0; s < a.nseg; ++s)
        if (a.seg[s].type != MI355_GGML_Q4_K && a.seg[s].type != MI355_GGML_Q6_K) return (int)hipErrorNotSupported;
    if (a.moe_expert || (a.K & 255)) return (int)hipErrorNotSupported;
    int n_slots = 0;
    for (int s = 0; s < a.nseg; ++s) n_slots += a.seg[s].n_tiles;
    const int T = a.B, K = a.K, ldp = n_slots * 16, nkb = K >> 8;
    const int Tpad = (T + QPG_BM - 1) / QPG_BM * QPG_BM;
    const int parts = g_tune_exact_act ? 2 : 1;             // activation planes of the f16 image (mi355_set_tuning(24, 1) = hi + lo)
    // (a block-table launch only exists with the epilogue in the store loop: no C buffer -- it would be 1.9 GB for a Mixtral chunk)
    const size_t xa_b = (size_t)K * Tpad * 2 * parts, sf_b = (size_t)nkb * Tpad * 32, rs_b = (size_t)Tpad * 4, c_b = a.blk_tab ? 0 : (size_t)T * ldp * 4;
    void* ws = nullptr;
    int rc = qmg_buf(&ws, MI355_SCR_QMP_WS, xa_b + sf_b + 2 * rs_b + c_b + 4096, st);
    if (rc) return rc;
    a.dbg = g_tune_dbg;                                     // ablation bits of the prompt-step GEMM (experiments only)
    uint8_t* base = static_cast<uint8_t*>(ws);
    QpgImg im;
    im.xa = base; im.sfrag = base + xa_b;
    im.row_scale = reinterpret_cast<float*>(base + xa_b + sf_b);
    im.row_inv = im.row_scale + Tpad;
    im.Tpad = Tpad;
    im.parts = parts;
    float* C = reinterpret_cast<float*>(base + xa_b + sf_b + 2 * rs_b);
    static Mi355DevOnce attr_done;
    if (!attr_done.done()) {
#define QPG_ATTR(MTW, NTW, WM, WN, DEEP) \
        (void)hipFuncSetAttribute((const void*)qpg_gemm_kernel<MI355_GGML_Q4_K, MTW, NTW, WM, WN, DEEP, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024); \
        (void)hipFuncSetAttribute((const void*)qpg_gemm_kernel<MI355_GGML_Q4_K, MTW, NTW, WM, WN, DEEP, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024)
        QPG_ATTR(4, 2, 1, 8, false);
#ifdef MI355_QMM_PROBES
        QPG_ATTR(2, 4, 1, 8, false); QPG_ATTR(4, 4, 1, 8, false); QPG_ATTR(2, 4, 2, 4, false);
        (void)hipFuncSetAttribute((const void*)qpg_gemm_kernel<MI355_GGML_Q4_K, 2, 4, 1, 8, false, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        (void)hipFuncSetAttribute((const void*)qpg_gemm_q6k_kernel<1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
#endif
#undef QPG_ATTR
        (void)hipFuncSetAttribute((const void*)qpg_gemm_lds2_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        (void)hipFuncSetAttribute((const void*)qpg_gemm_lds2_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        (void)hipFuncSetAttribute((const void*)qpg_gemm_lds2_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        (void)hipFuncSetAttribute((const void*)qpg_gemm_q6k_kernel<2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        (void)hipFuncSetAttribute((const void*)qpg_gemm_q6k_lds_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)qpg_gemm_q6k_lds_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)qpg_gemm_q6k_lds_kernel<true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)qpg_gemm_q6k_lds_kernel<true, 2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done.set();
    }
    if (a.blk_tab && g_tune_qpg != 2) return (int)hipErrorNotSupported;                   // only the LDS-fed kernel reads a block table
    hipLaunchKernelGGL(qpg_rowprep_kernel, dim3(Tpad), dim3(256), 0, st, a, im);          // row statistics + image, one launch (workgroup = token)
    // EXPERIMENT (mi355_set_tuning(48, 1)): all segments Q4_K, one activation plane, the default wave tile, and an epilogue that is a
    // store / residual add / SiLU * up over two equal segments -> the GEMM writes the outputs itself, no C buffer, no epilogue launch
    {
        bool all_q4 = true;
        for (int s = 0; s < a.nseg; ++s) all_q4 = all_q4 && a.seg[s].type == MI355_GGML_Q4_K;
        const bool epi_ok = a.epi == MI355_EPI_STORE || a.epi == MI355_EPI_RESID ||
                            (a.epi == MI355_EPI_SILU_MUL && a.nseg == 2 && a.seg[0].n_tiles == a.seg[1].n_tiles && a.seg[0].n_rows == a.seg[1].n_rows);
        if (g_tune_qpg_fepi && all_q4 && epi_ok && parts == 1 && (g_tune_qpg == 0 || g_tune_qpg == 2)) {
            QmmArgs r = a;
            r.norm_w = nullptr;                                 // applied while the image was built
            if (g_tune_qpg == 2) {                              // 64 tokens x 256 rows per workgroup (waves 64 x 32), two workgroups per CU
                const dim3 g_(Tpad / 64, (n_slots + 15) / 16), b_(512);
                if (a.blk_tab) hipLaunchKernelGGL((qpg_gemm_lds2_kernel<true, true>), g_, b_, (size_t)QPG2_BYTES, st, r, im, C, ldp, n_slots, 0);
                else hipLaunchKernelGGL((qpg_gemm_lds2_kernel<true>), g_, b_, (size_t)QPG2_BYTES, st, r, im, C, ldp, n_slots, 0);
            } else {
#ifdef MI355_QMM_PROBES
                const dim3 g_(Tpad / 32, (n_slots + 31) / 32), b_(512);   // rounds 2-3: 32 x 512, register-fed
                const size_t sh_ = (size_t)2 * 32 * QPG_ROWB;
                hipLaunchKernelGGL((qpg_gemm_kernel<MI355_GGML_Q4_K, 2, 4, 1, 8, false, 1, true>), g_, b_, sh_, st, r, im, C, ldp, n_slots, 0);
#endif
            }
            return (int)hipGetLastError();
        }
        if (a.blk_tab) {
            // block-table launches: Q4_K above; one Q6_K matrix with a store / residual epilogue here (the down projections of a Q4_K_M file)
            if (a.nseg == 1 && a.seg[0].type == MI355_GGML_Q6_K && (a.epi == MI355_EPI_STORE || a.epi == MI355_EPI_RESID) && parts == 1 &&
                g_tune_qpg == 2 && g_tune_qpg_fepi) {
                QmmArgs r = a;
                r.norm_w = nullptr;
                hipLaunchKernelGGL((qpg_gemm_q6k_lds_kernel<true, 2, true>), dim3(Tpad / 64, (n_slots + 15) / 16), dim3(512), (size_t)QPG6_LDS_BYTES, st, r, im, C,
                                   ldp, n_slots, 0);
                return (int)hipGetLastError();
            }
            return (int)hipErrorNotSupported;
        }
    }
    // q | k | v of a prompt step (EPI_QKV_ROPE_CACHE, round 6): q and k (Q4_K) -- and v where it is Q4_K too -- go through the fused store loop
    // (RoPE + bf16 + cache scatter in the GEMM); a Q6_K v (half the layers of a Q4_K_M file) keeps its own GEMM + the epilogue launch, which
    // then sees q and k as empty segments (n_rows = 0: their columns of C were never written)
#ifndef QPG_FUSE_Q6K
#define QPG_FUSE_Q6K 1          // A/B builds: 0 = Q6_K prompt GEMMs write the C buffer and an epilogue launch follows (rounds 2-5)
#endif
#ifndef QPG6_SPLIT_SMALL
#define QPG6_SPLIT_SMALL 1      // A/B builds: 0 = the Q6_K value projection always in 256-row workgroups
#endif
#ifndef QPG_FUSE_QKV
#define QPG_FUSE_QKV 1          // A/B builds: 0 = q | k | v through the C buffer and the epilogue launch (rounds 2-5)
#endif
    // (from 1024 tokens: measured +0.7 % at 2048 and +1.2 % at 4096 tokens, -1.9 % at 512 -- profiles/r06_prompt_qkv_fused_ab.txt)
    if (QPG_FUSE_QKV && T >= 1024 && g_tune_qpg_fepi && parts == 1 && g_tune_qpg == 2 && a.epi == MI355_EPI_QKV_ROPE_CACHE && a.nseg == 3 &&
        a.seg[0].type == MI355_GGML_Q4_K && a.seg[1].type == MI355_GGML_Q4_K && (QPG_FUSE_Q6K || a.seg[2].type == MI355_GGML_Q4_K)) {
        const bool v4 = a.seg[2].type == MI355_GGML_Q4_K;
        QmmArgs r = a;
        r.norm_w = nullptr;                                         // applied while the image was built
        if (!v4) r.nseg = 2;
        int f_slots = 0;
        for (int q = 0; q < r.nseg; ++q) f_slots += r.seg[q].n_tiles;
        hipLaunchKernelGGL((qpg_gemm_lds2_kernel<true>), dim3(Tpad / 64, (f_slots + 15) / 16), dim3(512), (size_t)QPG2_BYTES, st, r, im, C, ldp, f_slots, 0);
        if (v4) return (int)hipGetLastError();
        // v (Q6_K): its own GEMM with the fused store loop; q and k stay in the descriptor as EMPTY segments (no tiles) so that the value
        // segment keeps its role (2 = v: bf16 + cache scatter, no rotation) and its row offsets
        QmmArgs rv = a;
        rv.norm_w = nullptr;
        rv.seg[0].n_tiles = 0; rv.seg[1].n_tiles = 0;
        // (one row tile per wave where two would leave CUs empty: 1024 rows x 2048 tokens = 128 workgroups of 256 rows)
        const int vt = a.seg[2].n_tiles;
        if (QPG6_SPLIT_SMALL && (Tpad / 64) * ((vt + 15) / 16) < mi355_num_cus())
            hipLaunchKernelGGL((qpg_gemm_q6k_lds_kernel<true, 1>), dim3(Tpad / 64, (vt + 7) / 8), dim3(512), (size_t)QPG6_LDS_BYTES, st, rv, im, C, ldp, vt, 0);
        else
            hipLaunchKernelGGL((qpg_gemm_q6k_lds_kernel<true>), dim3(Tpad / 64, (vt + 15) / 16), dim3(512), (size_t)QPG6_LDS_BYTES, st, rv, im, C, ldp, vt, 0);
        return (int)hipGetLastError();
    }
    for (int s0 = 0, slot_base = 0; s0 < a.nseg;) {
        int s1 = s0 + 1;
        while (s1 < a.nseg && a.seg[s1].type == a.seg[s0].type) ++s1;
        QmmArgs r = a;
        r.nseg = s1 - s0;
        int run_slots = 0;
        for (int q = 0; q < r.nseg; ++q) { r.seg[q] = a.seg[s0 + q]; run_slots += r.seg[q].n_tiles; }
        if (r.seg[0].type == MI355_GGML_Q4_K) {
            // variants (mi355_set_tuning(11, v)): wave tile (m-tiles x row tiles) and wave grid; PARTS = activation planes
#define QPG_GO(MTW, NTW, WM, WN, DEEP) do { \
                const dim3 g_(Tpad / (16 * MTW * WM), (run_slots + NTW * WN - 1) / (NTW * WN)), b_(WM * WN * 64); \
                const size_t sh_ = (size_t)2 * parts * (16 * MTW * WM) * QPG_ROWB; \
                if (parts == 1) hipLaunchKernelGGL((qpg_gemm_kernel<MI355_GGML_Q4_K, MTW, NTW, WM, WN, DEEP, 1>), g_, b_, sh_, st, r, im, C, ldp, run_slots, slot_base); \
                else hipLaunchKernelGGL((qpg_gemm_kernel<MI355_GGML_Q4_K, MTW, NTW, WM, WN, DEEP, 2>), g_, b_, sh_, st, r, im, C, ldp, run_slots, slot_base); } while (0)
            switch (g_tune_qpg) {
#ifdef MI355_QMM_PROBES
                case 1: QPG_GO(4, 4, 1, 8, false); break;              //  64 x 512, waves 64 x 64
                case 3: QPG_GO(2, 4, 2, 4, false); break;              //  64 x 256, waves 32 x 64
                case 0: QPG_GO(2, 4, 1, 8, false); break;              //  32 x 512, waves 32 x 64: measured best in round 2 (hi + lo planes); with one plane 64 x 256 wins (round 4: 25.7 k -> 27.0 k tok/s at T = 2048)
#endif
                default:                                                //  64 x 256, waves 64 x 32, two workgroups per CU; one activation plane: image and weights through LDS by DMA
                    if (parts == 1) {
                        const dim3 g4(Tpad / 64, (run_slots + 15) / 16), b_(512);
                        hipLaunchKernelGGL((qpg_gemm_lds2_kernel<false>), g4, b_, (size_t)QPG2_BYTES, st, r, im, C, ldp, run_slots, slot_base);
                    } else {
                        QPG_GO(4, 2, 1, 8, false);                      // hi + lo planes ("exact" activations): the register-fed form
                    }
                    break;
            }
#undef QPG_GO
        } else {
            if (parts == 1 && g_tune_qpg == 2) {                        // Q6_K: 64 tokens x 256 rows per workgroup; token blocks fastest
                const dim3 grid(Tpad / 64, (run_slots + 15) / 16);
                // the whole mat-mul is this one Q6_K run and its epilogue is a store / residual add: fused store loop, no C buffer, no launch
                // (from 4096 tokens -- measured: +1.2 % at 4096, -0.2 % at 2048: profiles/r06_prompt_q6k_fused_ab.txt)
                if (QPG_FUSE_Q6K && g_tune_qpg_fepi && T >= 4096 && s0 == 0 && s1 == a.nseg && (a.epi == MI355_EPI_STORE || a.epi == MI355_EPI_RESID)) {
                    r.norm_w = nullptr;
                    hipLaunchKernelGGL((qpg_gemm_q6k_lds_kernel<true>), grid, dim3(512), (size_t)QPG6_LDS_BYTES, st, r, im, C, ldp, run_slots, 0);
                    return (int)hipGetLastError();
                }
                hipLaunchKernelGGL((qpg_gemm_q6k_lds_kernel<false>), grid, dim3(512), (size_t)QPG6_LDS_BYTES, st, r, im, C, ldp, run_slots, slot_base);
            } else {
                const dim3 grid(Tpad / 32, (run_slots + 15) / 16);     // 32 tokens x 256 rows, register-fed: hi + lo planes ("exact" activations)
#ifdef MI355_QMM_PROBES
                if (parts == 1) hipLaunchKernelGGL((qpg_gemm_q6k_kernel<1, 2>), grid, dim3(512), 8 * 1024, st, r, im, C, ldp, run_slots, slot_base);
                else
#endif
                hipLaunchKernelGGL((qpg_gemm_q6k_kernel<2, 2>), grid, dim3(512), 16 * 1024, st, r, im, C, ldp, run_slots, slot_base);
            }
        }
        slot_base += run_slots;
        s0 = s1;
    }
    a.norm_w = nullptr;                                     // applied while the image was built
    hipLaunchKernelGGL(qmm_epilogue_kernel, dim3((ldp + 255) / 256, T), dim3(256), 0, st, a, C, ldp, 1, T, (const float*)nullptr,
                       QmgChainOut{nullptr, nullptr, nullptr, 0, 0, 0, 0}, (const float*)im.row_scale);
    return (int)hipGetLastError();
}

#ifdef MI355_QMM_PROBES   // probe builds only (tools/build_probe_lib.sh): launcher of the first-generation prompt path
#include "probes/qmp_launch.inc"
#endif

// ------------------------------------------------------------------------------------------------ launcher
void mi355_pa_set_fused(int v);
void mi355_pa_set_wpb(int v);
void mi355_pa_set_loop(int v);
void mi355_dense_set_small(int key, int v);
extern "C" void mi355_host_set_partition_override(int v);
extern "C" void mi355_host_set_moe_group(int v);
extern "C" void mi355_dense_set_tile(int v);
void mi355_prefill_set_fp8_generic(int v);
void mi355_prefill_set_lds(int v);
static int g_tune_actq8 = 0;                               // mi355_set_tuning(18, 1): EXPERIMENT, single-token launches quantise x to Q8_K (reference CPU numerics, O2)
static int g_tune_nw = 0, g_tune_r = 0;                   // 0 = heuristic; mi355_set_tuning (experiments only)
static int g_tune_prefill_gemm = 1;                        // 0 = always stream the quantised weights (experiments)
// ---- A/B switches.  The PRODUCT library honours exactly ten keys (documented in include/mi355_vllm.h; their defaults live in the table
// below, which is applied through the same setters at load time, so `mi355_get_tuning` always returns the live value of a product key);
// every other key belongs to probe builds (-DMI355_QMM_PROBES, tools/build_probe_lib.sh) and is ignored by the product library.
void mi355_prefill_set_fp8_generic(int v);
void mi355_prefill_set_lds(int v);
static int32_t g_tune_shadow[64];
static bool g_tune_shadow_set[64];
static const int32_t kProductTuning[][2] = {{3, 1}, {5, 0}, {6, 1}, {9, 1}, {24, 0}, {30, 0}, {41, 1}, {44, 1}, {47, 1}, {48, 1}};
static bool tuning_is_product_key(int32_t key) {
    for (const auto& kv : kProductTuning) if (kv[0] == key) return true;
    return false;
}
static void tuning_apply(int32_t key, int32_t value) {
    if (key == 3) { mi355_pa_set_fused(value & 15); mi355_pa_set_wpb(value >> 4); }   // merge form | 16 x partitions per workgroup (0 = auto)
    else if (key == 5) mi355_host_set_partition_override(value);
    else if (key == 6) g_tune_prefill_gemm = value;
    else if (key == 9) g_tune_chain = value;
    else if (key == 24) g_tune_exact_act = value;
    else if (key == 30) mi355_dense_set_small(30, value);           // bit mask of folded launches switched OFF (dense_gemv.hip)
    else if (key == 41) mi355_host_set_moe_group(value);
    else if (key == 44) mi355_pa_set_loop(value);
    else if (key == 47) { mi355_prefill_set_lds(value & 1); mi355_prefill_set_fp8_generic((value >> 1) & 1); }
    else if (key == 48) g_tune_qpg_fepi = value;
#ifdef MI355_QMM_PROBES
    else if (key == 0) g_tune_nw = value;
    else if (key == 1) g_tune_r = value;
    else if (key == 2) g_tune_dbg = value;
    else if (key == 10 && value != 0) g_tune_ks_target = value;     // > 0: (row tile x k-split) slots, < 0: workgroups
    else if (key == 11) g_tune_qpg = value;
    else if (key == 12 && value > 0) g_tune_qpg_min = value;
    else if (key == 14) g_tune_merge = value;
    else if (key == 15) g_tune_wide16 = value;
    else if (key == 16 && value > 0) g_tune_grp_ks_target = value;
    else if (key == 17 && value > 0) g_tune_ks_minkb = value;
    else if (key == 18) g_tune_actq8 = value;
    else if (key == 20) g_tune_qmv = value;
    else if (key == 21 && value > 0) g_tune_qmv_nc = value;
    else if (key == 22) g_tune_qmv_ring = value;
    else if (key == 23) g_tune_chain_b1 = value;
    else if (key >= 33 && key <= 38) mi355_dense_set_small(key, value);
    else if (key == 42) mi355_dense_set_tile(value);
    else if (key == 49) g_tune_wide_fuse = value;
#endif
}
namespace {
struct TuningInit {
    TuningInit() {
        for (const auto& kv : kProductTuning) { tuning_apply(kv[0], kv[1]); g_tune_shadow[kv[0]] = kv[1]; g_tune_shadow_set[kv[0]] = true; }
#ifdef MI355_QMM_PROBES
        // probe builds: every key has a live value from the start (the defaults of the variables the setters write; ADVICE r4: a scoped
        // set / restore of a never-set key used to "restore" 0, which several setters ignore and which switches others off)
        static const int32_t kProbeDefaults[][2] = {{0, 0}, {1, 0}, {2, 0}, {10, 1024}, {11, 2}, {12, 96}, {14, 1}, {15, 0}, {16, 4096}, {17, 2}, {18, 0},
                                                    {20, 0}, {21, 8}, {22, 0}, {23, 0}, {33, 0}, {34, 0}, {35, 0}, {36, 96}, {37, 0}, {38, 4},
                                                    {42, 1}, {49, 0}};
        for (const auto& kv : kProbeDefaults) { g_tune_shadow[kv[0]] = kv[1]; g_tune_shadow_set[kv[0]] = true; }
#endif
    }
} g_tuning_init;
}
/* 1 when this build honours `key` (the ten product keys; every key in probe builds) */
extern "C" int32_t mi355_tuning_supported(int32_t key) {
    if (key < 0 || key >= 64) return 0;
#ifdef MI355_QMM_PROBES
    return 1;
#else
    return tuning_is_product_key(key) ? 1 : 0;
#endif
}
extern "C" int32_t mi355_get_tuning(int32_t key) {
    if (key < 0 || key >= 64 || !g_tune_shadow_set[key]) return INT32_MIN;
    return g_tune_shadow[key];
}
extern "C" void mi355_set_tuning(int32_t key, int32_t value) {
    if (!mi355_tuning_supported(key)) return;                       // product build: a probe key is ignored (and stays "never set")
    g_tune_shadow[key] = value; g_tune_shadow_set[key] = true;
    tuning_apply(key, value);
}

static size_t qmm_lds_bytes(int BT, int R, int NW) {
    const size_t xw = (size_t)32 * 2 * BT * 16;
    return (size_t)NW * xw + (size_t)NW * R * BT * 16 * 4 + (size_t)NW * BT * 4 + 64;
}

static int g_num_cus = 0;

// The LDS-DMA loader / consumer engine for single-token launches and the chained persistent launch built on it are
// EXPERIMENTS that lost their A/B on the MI355X ("what was tried for batch 1 in round 3"): they are
// compiled into probe builds only (tools/build_probe_lib.sh); the product library keeps the entry points as refusals.
#ifdef MI355_QMM_PROBES
#include "probes/qmv_engine.inc"
#else
static int qmv_launch(const QmmArgs&, int, hipStream_t) { return (int)hipErrorNotSupported; }
extern "C" int mi355_qmv_error(int32_t* out_host, int32_t) { if (out_host) *out_host = 0; return 0; }
#endif

// Waves per workgroup: the largest NW in {8,4,2,1} for which every workgroup of the launch is resident at
// once (no second dispatch round => no tail), judged by the occupancy the runtime reports for this variant.
template <int BT, int R, int WT>
static int qmm_pick_nw(int n_wg, int nkb) {
    static int blocks_per_cu[4] = {-1, -1, -1, -1};       // NW = 8,4,2,1
    g_num_cus = mi355_num_cus();                         // per device (common.h)
    int pick = 1;
    for (int i = 0; i < 3; ++i) {                         // never below 2 waves: one wave per tile serialises K
        const int nw = 8 >> i;
        if (nw > nkb) continue;
        if (blocks_per_cu[i] < 0) {
            int nb = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)qmm_kernel<BT, R, WT, 0, 1>, 64 * nw,
                                                             qmm_lds_bytes(BT, R, nw)) != hipSuccess || nb < 1) nb = 1;
            blocks_per_cu[i] = nb;
        }
        pick = nw;
        if ((long)n_wg <= (long)blocks_per_cu[i] * g_num_cus) return nw;
    }
    // more workgroups than the chip holds even at 2 waves each: several rounds anyway -> 4 waves
    return nkb >= 4 ? 4 : pick;
}

template <int BT, int R, int WT>
static int qmm_launch_btrw(QmmArgs& a, int n_wg, int NW, hipStream_t st) {
    a.kch = 0;
    static Mi355DevOnce attr_done;
    if (!attr_done.done()) {
        (void)hipFuncSetAttribute((const void*)qmm_kernel<BT, R, WT, 0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)qmm_kernel<BT, R, WT, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)qmm_kernel<BT, R, WT, 1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)qmm_kernel<BT, R, WT, 1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done.set();
    }
    // (token, slot) pairs of a mixture-of-experts launch are workgroups too: all of them should be resident at once
    // the paired (gate / up) single-token launch runs TWO waves per workgroup: measured on the product build, alternated on one box
    // (profiles/r06_b1_gateup_waves_ab.txt): 2 waves 1.841 ms per step, 4 waves 1.862, 8 waves 1.875.  The occupancy rule below picked 2 while
    // the kernel needed 134 VGPRs and flipped to 4 when a register-allocation change brought it to 127 -- a choice that depends on the
    // allocator's mood is not a choice (QMM_NW_PAIRED: A/B builds)
#ifndef QMM_NW_PAIRED
#define QMM_NW_PAIRED 2
#endif
    if (NW <= 0 && a.paired && BT == 1 && !a.moe_expert && a.K / 256 >= QMM_NW_PAIRED) NW = QMM_NW_PAIRED;
    if (NW <= 0) NW = qmm_pick_nw<BT, R, WT>(n_wg * (a.moe_expert ? a.moe_pairs : 1), a.K / 256);
    const size_t shm = qmm_lds_bytes(BT, R, NW);
    if (shm > 160 * 1024) return (int)hipErrorInvalidValue;
    if (a.moe_expert) {
        if constexpr (BT == 1) {
            static Mi355DevOnce moe_attr_done;
            if (!moe_attr_done.done()) {
                (void)hipFuncSetAttribute((const void*)qmm_moe_kernel<R, WT, 0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                (void)hipFuncSetAttribute((const void*)qmm_moe_kernel<R, WT, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                (void)hipFuncSetAttribute((const void*)qmm_moe_kernel<R, WT, 1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                (void)hipFuncSetAttribute((const void*)qmm_moe_kernel<R, WT, 1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                moe_attr_done.set();
            }
            const bool xb = a.x_dtype == MI355_DTYPE_BF16, nrm = a.norm_w != nullptr;
            const dim3 mgrid(n_wg, a.moe_pairs), mblock(64 * NW);
            if (xb && nrm) hipLaunchKernelGGL((qmm_moe_kernel<R, WT, 1, 1>), mgrid, mblock, shm, st, a);
            else if (xb) hipLaunchKernelGGL((qmm_moe_kernel<R, WT, 1, 0>), mgrid, mblock, shm, st, a);
            else if (nrm) hipLaunchKernelGGL((qmm_moe_kernel<R, WT, 0, 1>), mgrid, mblock, shm, st, a);
            else hipLaunchKernelGGL((qmm_moe_kernel<R, WT, 0, 0>), mgrid, mblock, shm, st, a);
        } else {
            return (int)hipErrorInvalidValue;
        }
    } else {
#ifdef MI355_QMM_PROBES
        if constexpr (BT == 1) {
            if (g_tune_actq8) {
                const bool xb8 = a.x_dtype == MI355_DTYPE_BF16, nrm8 = a.norm_w != nullptr;
#define Q8_GO(XB_, NRM_) do { \
                    (void)hipFuncSetAttribute((const void*)qmm_q8_kernel<R, WT, XB_, NRM_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
                    hipLaunchKernelGGL((qmm_q8_kernel<R, WT, XB_, NRM_>), dim3(n_wg), dim3(64 * NW), shm, st, a); } while (0)
                if (xb8 && nrm8) Q8_GO(1, 1); else if (xb8) Q8_GO(1, 0); else if (nrm8) Q8_GO(0, 1); else Q8_GO(0, 0);
#undef Q8_GO
                return (int)hipGetLastError();
            }
        }
#endif
        const bool xb = a.x_dtype == MI355_DTYPE_BF16, nrm = a.norm_w != nullptr;
        if (xb && nrm) hipLaunchKernelGGL((qmm_kernel<BT, R, WT, 1, 1>), dim3(n_wg), dim3(64 * NW), shm, st, a);
        else if (xb) hipLaunchKernelGGL((qmm_kernel<BT, R, WT, 1, 0>), dim3(n_wg), dim3(64 * NW), shm, st, a);
        else if (nrm) hipLaunchKernelGGL((qmm_kernel<BT, R, WT, 0, 1>), dim3(n_wg), dim3(64 * NW), shm, st, a);
        else hipLaunchKernelGGL((qmm_kernel<BT, R, WT, 0, 0>), dim3(n_wg), dim3(64 * NW), shm, st, a);
    }
    return (int)hipGetLastError();
}

template <int BT, int R>
static int qmm_launch_btr(QmmArgs& a, int wt, int n_wg, int NW, hipStream_t st) {
    if (wt == MI355_GGML_Q4_K) return qmm_launch_btrw<BT, R, MI355_GGML_Q4_K>(a, n_wg, NW, st);
    if (wt == MI355_GGML_Q6_K) return qmm_launch_btrw<BT, R, MI355_GGML_Q6_K>(a, n_wg, NW, st);
    return qmm_launch_btrw<BT, R, 0>(a, n_wg, NW, st);
}

template <int BT>
static int qmm_launch_bt(QmmArgs& a, int R, int wt, int n_wg, int NW, hipStream_t st) {
    if (R == 4) return qmm_launch_btr<BT, 4>(a, wt, n_wg, NW, st);
    if (R == 2) return qmm_launch_btr<BT, 2>(a, wt, n_wg, NW, st);
    return qmm_launch_btr<BT, 1>(a, wt, n_wg, NW, st);
}

// Run one fused quantised mat-mul over batch rows [0,B) in M-tiles of 8 batch entries.
// ================================================================================================
// Q8_0 arm: tensor-parallel shards the reference re-quantises (dim-1 shards that cut a 256-wide k-quant block are
// dequantised, narrowed and re-quantised to Q8_0 -- layers/quantized_var_builder.rs:234-269).  Native blocks
// [rows][K/32][f16 d | 32 x i8], no repack; one wave per output row and up to 8 tokens, lanes stride over the blocks.
// A fallback for odd shapes (o_proj / down_proj of a model whose k / W is not a multiple of 256), not a tuned path.
__global__ void __launch_bounds__(64) q8_0_matmul_kernel(const QmmArgs a) {
    const int row = blockIdx.x, t0 = blockIdx.y * 8, lane = threadIdx.x;
    const int nt = min(8, a.B - t0), nblk = a.K / 32;
    const uint8_t* wr = a.seg[0].w + (size_t)row * nblk * 34;
    float acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = 0.f;
    for (int b = lane; b < nblk; b += 64) {
        const uint16_t* p16 = reinterpret_cast<const uint16_t*>(wr + (size_t)b * 34);      // 34-byte blocks: 2-byte aligned
        const float d = f16_bits_to_f32(p16[0]);
        float w[32];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const uint16_t u = p16[1 + j];
            w[2 * j] = d * (float)(int8_t)(u & 0xFF);
            w[2 * j + 1] = d * (float)(int8_t)(u >> 8);
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            if (t >= nt) break;
            float s = 0.f;
            if (a.x_dtype == MI355_DTYPE_BF16) {
                const uint16_t* xp = static_cast<const uint16_t*>(a.x) + (size_t)(t0 + t) * a.ldx + (size_t)b * 32;
#pragma unroll
                for (int j = 0; j < 32; ++j) s = fmaf(bf16_to_f32(xp[j]), w[j], s);
            } else {
                const float* xp = static_cast<const float*>(a.x) + (size_t)(t0 + t) * a.ldx + (size_t)b * 32;
#pragma unroll
                for (int j = 0; j < 32; ++j) s = fmaf(xp[j], w[j], s);
            }
            acc[t] += s;
        }
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        if (t >= nt) break;
        float v = wave_sum(acc[t]);
        if (lane == 0) {
            if (a.bias) v += a.bias[row];
            const size_t o = (size_t)(t0 + t) * a.ldo + row;
            a.out[o] = (a.epi == MI355_EPI_RESID) ? a.resid[o] + v : v;
        }
    }
}
static int q8_0_launch(const QmmArgs& a, hipStream_t st) {
    if (a.nseg != 1 || a.norm_w || a.moe_expert || (a.epi != MI355_EPI_STORE && a.epi != MI355_EPI_RESID) || (a.K % 32) || a.K <= 0)
        return (int)hipErrorInvalidValue;
    if (a.B == 0) return 0;
    hipLaunchKernelGGL(q8_0_matmul_kernel, dim3(a.seg[0].n_rows, (a.B + 7) / 8), dim3(64), 0, st, a);
    return (int)hipGetLastError();
}

// ---- parity mode 2 (tests only; kernel in qmm_exact.inc): switch + launcher
static int g_qmm_exact = 0;
extern "C" void mi355_internal_qmm_set_exact(int32_t on) {
    g_qmm_exact = on ? 1 : 0;
    // an exact launch stages no activation image for its successor: forget every chain hint, in both directions of the switch
    std::lock_guard<std::mutex> lk(g_qmg_mu);
    for (auto& kv : g_qmg_streams) kv.second.chain.valid = false;
}
extern "C" int32_t mi355_internal_qmm_get_exact(void) { return g_qmm_exact; }

// one launch per token
static int qmm_exact_launch(QmmArgs a, hipStream_t st) {
    const int B = a.B;
    const size_t xes = (a.x_dtype == MI355_DTYPE_BF16) ? 2 : 4;
    const uint8_t* x0 = static_cast<const uint8_t*>(a.x);
    float* out0 = a.out;
    const float* resid0 = a.resid;
    uint16_t* q0 = a.q_out;
    const int64_t* pos0 = a.positions;
    const int64_t* slot0 = a.slot_mapping;
    int total_tiles = 0;
    for (int s = 0; s < a.nseg; ++s) total_tiles += a.seg[s].n_tiles;
    for (int b = 0; b < B; ++b) {
        a.B = 1;
        a.x = x0 + (size_t)b * a.ldx * xes;
        a.out = out0 ? out0 + (size_t)b * a.ldo : nullptr;
        a.resid = resid0 ? resid0 + (size_t)b * a.ldo : nullptr;
        a.q_out = q0 ? q0 + (size_t)b * a.Hq * a.D : nullptr;
        a.positions = pos0 ? pos0 + b : nullptr;
        a.slot_mapping = slot0 ? slot0 + b : nullptr;
        if (a.paired) hipLaunchKernelGGL((qmm_exact_kernel<2>), dim3(a.seg[0].n_tiles), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((qmm_exact_kernel<1>), dim3(total_tiles), dim3(256), 0, st, a);
    }
    return (int)hipGetLastError();
}

static bool qmg_wide_one_launch(int B) {
    return B > 8 && B <= 8 * QMW_MAXMT && !g_tune_exact_act && !g_qmm_exact && !(B >= g_tune_qpg_min && g_tune_prefill_gemm);
}

int mi355_qmm_launch(QmmArgs a, int64_t stream) {
    {   // images staged WITHOUT backing activations (fused MoE staging) may only be consumed by the grouped 9..32-token launch they were
        // built for; any other route would read rows nobody wrote.  qw1_launch checks the remaining fields (tile height, k, norm, groups)
        QmgStream& qs0 = qmg_stream(to_stream(stream));
        if (qs0.chain.valid && qs0.chain.prestaged && qs0.chain.x == a.x && qs0.chain.st == to_stream(stream) &&
            !(a.grp_n > 1 && qmg_wide_one_launch(a.B) && a.nseg >= 1 && a.seg[0].type != MI355_GGML_Q8_0)) {
            qs0.chain.valid = false;
            mi355_note_error((int)hipErrorInvalidValue);
            return (int)hipErrorInvalidValue;
        }
    }
    if (a.nseg >= 1 && a.seg[0].type == MI355_GGML_Q8_0) return q8_0_launch(a, to_stream(stream));
    if (a.K <= 0 || (a.K % 256) || a.B < 0 || a.nseg < 1 || a.nseg > 3) return (int)hipErrorInvalidValue;
    if (a.B == 0) return 0;
    for (int s = 0; s < a.nseg; ++s)
        if (a.seg[s].type != MI355_GGML_Q4_K && a.seg[s].type != MI355_GGML_Q6_K) return (int)hipErrorInvalidValue;
    if (a.paired && (a.nseg != 2 || a.seg[0].n_tiles != a.seg[1].n_tiles)) return (int)hipErrorInvalidValue;
    if (a.grp_n > 1) {
        // grouped launch: grp_n mat-muls of these shapes.  The 9..32-token path runs them as the z extent of its launches; everything
        // else (other token counts, the exact modes) runs them one after the other -- same results, the descriptor's meaning is the loop
        if (a.moe_expert || a.bias || (a.epi != MI355_EPI_STORE && a.epi != MI355_EPI_SILU_MUL)) return (int)hipErrorInvalidValue;
        const bool one_launch = qmg_wide_one_launch(a.B);
        if (!one_launch) {
            const size_t xes = (a.x_dtype == MI355_DTYPE_BF16) ? 2 : 4;
            for (int e = 0; e < a.grp_n; ++e) {
                QmmArgs r = a;
                r.grp_n = 0; r.chain_next = 0;
                for (int s = 0; s < a.nseg; ++s) r.seg[s].w = a.seg[s].w + (size_t)e * a.moe_stride[s];
                r.x = static_cast<const uint8_t*>(a.x) + (size_t)e * a.grp_x * xes;
                r.out = a.out + (size_t)e * a.grp_out;
                if (a.rows_dev) r.rows_dev = a.rows_dev + e;
                const int rc = mi355_qmm_launch(r, stream);
                if (rc) return rc;
            }
            return 0;
        }
    }
    if (g_qmm_exact && !a.moe_expert && !a.rows_dev) return qmm_exact_launch(a, to_stream(stream));   // parity mode 2 (tests only)
    int total_tiles = 0;
    bool div4 = true, div2 = true;
    for (int s = 0; s < a.nseg; ++s) {
        total_tiles += a.seg[s].n_tiles;
        if (a.seg[s].n_tiles % 4) div4 = false;
        if (a.seg[s].n_tiles % 2) div2 = false;
    }
    // tiles per workgroup: amortise the activation staging but keep >= ~1024 workgroups (4 per CU)
    int R;
    if (a.paired) {
        R = (a.seg[0].n_tiles % 2 == 0 && a.seg[0].n_tiles / 2 >= 1024) ? 4 : 2;
    } else {
        R = 1;
        if (div2 && total_tiles / 2 >= 1024) R = 2;
        // four Q6_K tiles per workgroup need more than the register file (256 VGPRs + spills once the body is branch-free):
        // lm_head of Llama-3-8B 108 us with R = 4, 74 us with R = 2
        bool any_q6 = false;
        for (int s = 0; s < a.nseg; ++s) any_q6 = any_q6 || a.seg[s].type == MI355_GGML_Q6_K;
        if (div4 && total_tiles / 4 >= 1024 && !any_q6) R = 4;
    }
    if (g_tune_r > 0 && !(a.paired && g_tune_r == 1)) {
        if (g_tune_r == 4 && (a.paired ? a.seg[0].n_tiles % 2 == 0 : div4)) R = 4;
        else if (g_tune_r == 2 && (a.paired || div2)) R = 2;
        else if (g_tune_r == 1 && !a.paired) R = 1;
    }
    const int n_wg = a.paired ? a.seg[0].n_tiles / (R / 2) : total_tiles / R;
    const int nkb = a.K / 256;
    int NW = 0;                                               // 0 = occupancy-aware choice per kernel variant
    if (g_tune_nw > 0) { NW = g_tune_nw; while (NW > 1 && nkb < NW) NW >>= 1; if (NW > 8) NW = 8; }
    hipStream_t st = to_stream(stream);
    int wt = a.seg[0].type;                                   // uniform tile type of the launch, else 0 (mixed)
    for (int s = 1; s < a.nseg; ++s) if (a.seg[s].type != wt) wt = 0;
    a.dbg = g_tune_dbg;
    if (a.moe_expert) {                                       // every (token, slot) pair is a 1-token mat-vec of its own expert
        if (a.moe_pairs < 1 || a.moe_xdiv < 1 || a.epi == MI355_EPI_QKV_ROPE_CACHE || a.epi == MI355_EPI_RESID) return (int)hipErrorInvalidValue;
        a.B = 1;
        return qmm_launch_bt<1>(a, R, wt, n_wg, NW, st);
    }
    if (a.blk_tab && !(a.B >= g_tune_qpg_min && g_tune_prefill_gemm == 1)) return (int)hipErrorNotSupported;   // only the prompt GEMM reads a block table
    if (a.B >= g_tune_qpg_min && g_tune_prefill_gemm) {
        // prompt step: the hand-written quantised GEMM (qmm_prefill.inc).  mi355_set_tuning(6, 2) = the first-generation path
        // (bf16 hi/lo weight image + three library GEMMs) for A/B runs; (6, 0) = stream the weights 32 tokens at a time.
#ifdef MI355_QMM_PROBES
        if (g_tune_prefill_gemm == 2) {
            const int rcp = qmp_launch(a, st);
            if (rcp != (int)hipErrorSharedObjectInitFailed) return rcp;
        } else
#endif
        {
            const int rcq = qpg_launch(a, st);
            if (rcq != (int)hipErrorNotSupported || a.blk_tab) return rcq;
        }
    }
    const int B = a.B;
    const size_t xes = (a.x_dtype == MI355_DTYPE_BF16) ? 2 : 4;
    const uint8_t* x0 = static_cast<const uint8_t*>(a.x);
    float* out0 = a.out;
    const float* resid0 = a.resid;
    uint16_t* q0 = a.q_out;
    const int64_t* pos0 = a.positions;
    const int64_t* slot0 = a.slot_mapping;
    for (int b0 = 0; b0 < B;) {
        int bn = B - b0;
        if (bn > 8) {                                         // 9..32 tokens: wide kernel, weights read once
            if (bn > 8 * QMW_MAXMT) bn = 8 * QMW_MAXMT;
            a.B = bn;
            a.x = x0 + (size_t)b0 * a.ldx * xes;
            a.out = out0 ? out0 + (size_t)b0 * a.ldo : nullptr;
            a.resid = resid0 ? resid0 + (size_t)b0 * a.ldo : nullptr;
            a.q_out = q0 ? q0 + (size_t)b0 * a.Hq * a.D : nullptr;
            a.positions = pos0 ? pos0 + b0 : nullptr;
            a.slot_mapping = slot0 ? slot0 + b0 : nullptr;
            int rcw;
            if (!g_tune_exact_act) rcw = bn <= 16 ? qw1_launch<1>(a, st) : qw1_launch<2>(a, st);   // one f16 plane, 16-token m-tiles (default)
            else if (bn <= 16) rcw = qmg_launch<2>(a, st);                                          // hi + lo planes, 8-token m-tiles ("exact")
            else if (bn <= 24) rcw = qmg_launch<3>(a, st);
            else rcw = qmg_launch<4>(a, st);
            if (rcw) return rcw;
            b0 += bn;
            continue;
        }
        a.B = bn;
        a.x = x0 + (size_t)b0 * a.ldx * xes;
        a.out = out0 ? out0 + (size_t)b0 * a.ldo : nullptr;
        a.resid = resid0 ? resid0 + (size_t)b0 * a.ldo : nullptr;
        a.q_out = q0 ? q0 + (size_t)b0 * a.Hq * a.D : nullptr;
        a.positions = pos0 ? pos0 + b0 : nullptr;
        a.slot_mapping = slot0 ? slot0 + b0 : nullptr;
        int rc;
        if (bn == 1) {
            rc = qmv_launch(a, wt, st);                       // LDS-DMA loader / consumer engine (qmv_engine.inc)
            if (rc == (int)hipErrorNotSupported) rc = qmm_launch_bt<1>(a, R, wt, n_wg, NW, st);
        }
        else if (bn == 2) rc = qmm_launch_bt<2>(a, R, wt, n_wg, NW, st);
        else if (bn <= 4) rc = qmm_launch_bt<4>(a, R, wt, n_wg, NW, st);
        else rc = qmm_launch_bt<8>(a, R, wt, n_wg, NW, st);
        if (rc) return rc;
        b0 += bn;
    }
    return 0;
}

/* internal (host_model.cpp, grouped experts of a decode step): route ids -> the grouped activation images of the NEXT grouped
 * mi355_qmatmul_fused call on this stream (x = x_key, `rows` tokens, k = hidden, the same norm weight, group_count = n_expert,
 * group_x_stride = cap * hidden), which finds them through the chain state and skips its staging launch; also leaves pos[] and counts[]
 * (mi355_moe_group's outputs).  -4: this configuration does not take the one-launch grouped path -- the caller groups and gathers. */
extern "C" int mi355_internal_moe_stage_grouped(const float* xs, const int32_t* ids, int32_t pairs, int32_t top_k, int32_t n_expert, int32_t cap,
                                                int32_t r0, int32_t rows, int32_t hidden, const float* norm_w, int32_t* pos_out,
                                                int32_t* counts_out, const float* x_key, int32_t row_limit, int64_t stream) {
    if (!xs || !ids || !pos_out || !counts_out || pairs < 1 || top_k < 1 || n_expert < 2 || cap < pairs || (hidden % 256)) return -4;
    if (!qmg_wide_one_launch(rows)) return -4;                          // the consumer's own path decision (shared helper)
    hipStream_t st = to_stream(stream);
    const int MT = rows <= 16 ? 1 : 2, BP = MT * 16, nkb = hidden / 256;
    const size_t kbb = qw1_kb_bytes(MT);
    QmgStream& qs = qmg_stream(st);
    qs.chain.valid = false;
    const int cur = qs.cur;
    const size_t imgb = (kbb * nkb + (size_t)nkb * BP * sizeof(float) + 1023) / 1024 * 1024;
    void* imgp = nullptr;
    const int rc = qmg_buf(&imgp, cur ? MI355_SCR_QMM_IMG1 : MI355_SCR_QMM_IMG0, (size_t)n_expert * imgb, st);
    if (rc) return rc;
    uint8_t* img = static_cast<uint8_t*>(imgp);
    float* ssp = reinterpret_cast<float*>(img + kbb * nkb);
    if (MT == 1)
        hipLaunchKernelGGL((qw1_prep_moe_kernel<1>), dim3(nkb, 2, n_expert), dim3(256), 0, st, img, ssp, xs, ids, pairs, top_k, n_expert, cap, r0,
                           hidden, norm_w, kbb, (int64_t)imgb, pos_out, counts_out, row_limit > 0 ? row_limit : cap);
    else
        hipLaunchKernelGGL((qw1_prep_moe_kernel<2>), dim3(nkb, 4, n_expert), dim3(256), 0, st, img, ssp, xs, ids, pairs, top_k, n_expert, cap, r0,
                           hidden, norm_w, kbb, (int64_t)imgb, pos_out, counts_out, row_limit > 0 ? row_limit : cap);
    qs.chain = QmgChainState{true, x_key, rows, hidden, MT, cur, norm_w, st, 1};
    qs.chain.grp = n_expert; qs.chain.grp_stride = (int64_t)cap * hidden;
    qs.chain.prestaged = true;                                          // no activations behind x_key: consume these images or fail
    return (int)hipGetLastError();
}

// The weighted expert rows added to the residual stream (mi355_moe_scatter_combine's arithmetic, bit for bit) AND the activation image of
// the next 9..32-token mat-mul that reads the residual stream (the next layer's q|k|v or the lm_head; its RMSNorm weight = next_norm_w):
// workgroup (256 columns, token row) = one (row, k-block) of the image, as in the chained epilogue.  grid.y covers the padded rows.
__global__ void __launch_bounds__(256) moe_scatter_combine_img_kernel(float* __restrict__ ys, const float* __restrict__ yg, const float* __restrict__ wts,
                                                                      const int32_t* __restrict__ inv, const int hidden, const int K, const int B,
                                                                      const QmgChainOut ch) {
    const int t = blockIdx.y;
    const int i = blockIdx.x * 256 + (int)threadIdx.x;
    float acc = 0.f;
    if (t < B) {
        acc = ys[(size_t)t * hidden + i];                               // residual (quantized_llama.rs:470)
        for (int j = 0; j < K; ++j) acc = fmaf(wts[(size_t)t * K + j], yg[(size_t)inv[t * K + j] * hidden + i], acc);
        ys[(size_t)t * hidden + i] = acc;
    }
    __shared__ float sm_o[256];
    sm_o[threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.x < 32) {
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = sm_o[threadIdx.x * 8 + q];
        qw1_prep_entry(ch.img, ch.ssp, v, t < B, ch.norm_w, ch.MT, ch.kbb, (int)blockIdx.x, t, (int)threadIdx.x);
    }
}
/* internal (host_model.cpp): -4 = this configuration does not chain (the caller runs mi355_moe_scatter_combine) */
extern "C" int mi355_internal_moe_scatter_combine_to_image(float* ys, const float* y_rows, const float* weights, const int32_t* inv, int32_t num_tokens,
                                                           int32_t hidden, int32_t top_k, const float* next_norm_w, int64_t stream) {
    if (!ys || !y_rows || !weights || !inv || top_k < 1 || (hidden % 256)) return -4;
    if (!g_tune_chain || !qmg_wide_one_launch(num_tokens)) return -4;
    hipStream_t st = to_stream(stream);
    const int MT = num_tokens <= 16 ? 1 : 2, BP = MT * 16, nkb = hidden / 256;
    const size_t kbb = qw1_kb_bytes(MT);
    QmgStream& qs = qmg_stream(st);
    qs.chain.valid = false;
    const int other = qs.cur ^ 1;
    void* imgp = nullptr;
    const int rc = qmg_buf(&imgp, other ? MI355_SCR_QMM_IMG1 : MI355_SCR_QMM_IMG0, kbb * nkb + (size_t)nkb * BP * sizeof(float), st);
    if (rc) return rc;
    uint8_t* img = static_cast<uint8_t*>(imgp);
    const QmgChainOut ch{img, reinterpret_cast<float*>(img + kbb * nkb), next_norm_w, hidden, MT, kbb, 1};
    hipLaunchKernelGGL(moe_scatter_combine_img_kernel, dim3(nkb, BP), dim3(256), 0, st, ys, y_rows, weights, inv, hidden, top_k, num_tokens, ch);
    qs.chain = QmgChainState{true, ys, num_tokens, hidden, MT, other, next_norm_w, st, 1};
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ C ABI
extern "C" int mi355_qmatmul(float* out, const float* x, const void* w_tiles, int32_t ggml_type, int32_t T,
                             int32_t N, int32_t K, const float* bias, int64_t stream) {
    if (N <= 0) return (int)hipErrorInvalidValue;
    QmmArgs a{};
    a.seg[0] = QmmSeg{static_cast<const uint8_t*>(w_tiles), ggml_type, (N + 15) / 16, 0, N};
    a.nseg = 1;
    a.x = x; a.x_dtype = MI355_DTYPE_F32; a.ldx = K; a.K = K; a.B = T;
    a.epi = MI355_EPI_STORE; a.out = out; a.ldo = N; a.bias = bias;
    return mi355_qmm_launch(a, stream);
}

static int qmm_args_from_desc(const mi355_qmm_desc* d, QmmArgs& a) {
    if (!d) return (int)hipErrorInvalidValue;
    a = QmmArgs{};
    a.nseg = d->nseg;
    if (d->nseg < 1 || d->nseg > 3) return (int)hipErrorInvalidValue;
    int row0 = 0;
    for (int s = 0; s < d->nseg; ++s) {
        a.seg[s] = QmmSeg{static_cast<const uint8_t*>(d->w_tiles[s]), d->ggml_type[s], (d->n_rows[s] + 15) / 16, row0, d->n_rows[s]};
        row0 += d->n_rows[s];
    }
    a.paired = (d->epilogue == MI355_EPI_SILU_MUL);
    if (a.paired) { a.seg[1].row0 = a.seg[0].n_rows; }
    a.x = d->x; a.x_dtype = d->x_dtype; a.ldx = d->ldx; a.K = d->k; a.B = d->num_tokens;
    if (a.x_dtype != MI355_DTYPE_F32 && a.x_dtype != MI355_DTYPE_BF16) return (int)hipErrorInvalidValue;
    a.norm_w = d->norm_weight; a.eps = d->norm_eps;
    a.epi = d->epilogue; a.out = d->out; a.ldo = d->ldo; a.resid = d->residual; a.bias = d->bias;
    a.cos_t = d->cos_table; a.sin_t = d->sin_table; a.positions = d->positions; a.slot_mapping = d->slot_mapping;
    a.q_out = static_cast<uint16_t*>(d->q_out); a.kcache = static_cast<uint16_t*>(d->key_cache);
    a.vcache = static_cast<uint16_t*>(d->value_cache);
    a.Hq = d->num_heads; a.Hkv = d->num_kv_heads; a.D = d->head_dim; a.rot = d->rotary_dim;
    a.block_size = d->block_size; a.kv_layout = d->kv_layout;
    if (a.epi == MI355_EPI_QKV_ROPE_CACHE) {
        if (d->nseg != 3 || !a.q_out || !a.kcache || !a.vcache || !a.cos_t || !a.sin_t || !a.positions ||
            !a.slot_mapping || a.D <= 0 || (a.D & 1) || a.rot <= 0 || (a.rot & 1) || a.rot > a.D ||
            d->n_rows[0] != a.Hq * a.D || d->n_rows[1] != a.Hkv * a.D || d->n_rows[2] != a.Hkv * a.D ||
            (d->n_rows[0] % 16) || (d->n_rows[1] % 16))
            return (int)hipErrorInvalidValue;
    }
    if (a.epi == MI355_EPI_RESID && !a.resid) return (int)hipErrorInvalidValue;
    if ((a.epi == MI355_EPI_STORE || a.epi == MI355_EPI_RESID || a.epi == MI355_EPI_SILU_MUL) && !a.out)
        return (int)hipErrorInvalidValue;
    a.moe_expert = d->moe_expert_ids; a.moe_pairs = d->moe_pairs; a.moe_xdiv = d->moe_x_div;
    a.rows_dev = d->rows_dev; a.rows_min = d->rows_min;
    for (int s = 0; s < 3; ++s) a.moe_stride[s] = d->moe_expert_stride[s];
    if (d->group_count < 0 || (d->group_count > 1 && (d->moe_expert_ids || d->group_x_stride < 0 || d->group_out_stride < 0))) return (int)hipErrorInvalidValue;
    a.grp_n = d->group_count > 1 ? d->group_count : 0; a.grp_x = d->group_x_stride; a.grp_out = d->group_out_stride;
    if (d->group_block_table && (d->moe_expert_ids || d->group_count > 1 || (d->num_tokens % 64))) return (int)hipErrorInvalidValue;
    a.blk_tab = d->group_block_table;
    a.chain_next = (d->chain_next && !d->moe_expert_ids && d->num_tokens > 8 && d->num_tokens <= 8 * QMW_MAXMT) ? 1 : 0;
    a.next_k = d->chain_next_k; a.next_norm_w = d->chain_next_norm;
    return 0;
}

extern "C" int mi355_qmatmul_fused(const mi355_qmm_desc* d, int64_t stream) {
    QmmArgs a{};
    const int rc = qmm_args_from_desc(d, a);
    if (rc) return rc;
    return mi355_qmm_launch(a, stream);
}

#ifdef MI355_QMM_PROBES
#include "probes/qmv_chain.inc"
#else
extern "C" int mi355_qmv_chain_sync_bytes(void) { return 256; }
extern "C" int mi355_qmatmul_chain(const mi355_qmm_desc*, int32_t, void*, int64_t) { return (int)hipErrorNotSupported; }
#endif
