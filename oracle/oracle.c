/* CPU oracle, C twin (TEST INFRASTRUCTURE ONLY -- never linked into the product path).
 *
 * A plain-C restatement of the reference's CPU arithmetic for the decode hot path, used (a) as a second,
 * independent statement of the numpy oracle (oracle/kquants.py, oracle/ops.py, oracle/llama.py) and
 * (b) as the `cpu_baseline` of bench.py (kind "port": the reference binary itself cannot be built here --
 * no Rust toolchain, and its arithmetic lives in un-vendored git dependencies).
 *
 * PARITY UNPINNED: no reference fixture exists for this path.  What is restated, with the call sites:
 *   QMatMul::forward on CPU = quantise activations per 256 to Q8_K, integer vec-dot   [EXT candle k_quants.rs]
 *       called from src/openai/models/layers/attention.rs:920-922,1004 ; quantized_llama.rs:33-37
 *   candle_nn::ops::rms_norm            src/openai/models/layers/qrmsnorm.rs:28-31
 *   candle rope_i (interleaved)         src/openai/models/layers/rotary_emb.rs:72-100, quantized_llama.rs:313-318
 *   NaiveAttention math                 src/openai/models/mod.rs:1288-1306 (GQA repeat_kv :1240-1247)
 *   layer loop / residuals / lm_head    src/openai/models/quantized_llama.rs:424-506
 *   slot / block table semantics        src/openai/pipelines/inputs.rs:376-454
 * Build: gcc -O3 -march=x86-64-v3 -fopenmp -shared -fPIC oracle/oracle.c -o oracle/liboracle.so -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define QK_K 256
#define Q4K_BYTES 144
#define Q6K_BYTES 210
#define T_Q4K 12
#define T_Q6K 14

/* ---- f16 / bf16 helpers ---------------------------------------------------------------------- */
static float f16_to_f32(uint16_t h) {
    const uint32_t s = (uint32_t)(h & 0x8000) << 16;
    uint32_t e = (h >> 10) & 0x1F, m = h & 0x3FF, u;
    if (e == 0) {
        if (m == 0) u = s;
        else {
            e = 127 - 15 + 1;
            while (!(m & 0x400)) { m <<= 1; --e; }
            u = s | (e << 23) | ((m & 0x3FF) << 13);
        }
    } else if (e == 31) u = s | 0x7F800000u | (m << 13);
    else u = s | ((e + 127 - 15) << 23) | (m << 13);
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static uint16_t f32_to_bf16(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return 0x7FC0;
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static float bf16_to_f32(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static float round_bf16(float f) { return bf16_to_f32(f32_to_bf16(f)); }

/* ---- k-quant formats ------------------------------------------------------- */
static void scale_min_k4(int j, const uint8_t* q, uint8_t* d, uint8_t* m) {
    if (j < 4) { *d = q[j] & 63; *m = q[j + 4] & 63; }
    else { *d = (q[j + 4] & 0xF) | ((q[j - 4] >> 6) << 4); *m = (q[j + 4] >> 4) | ((q[j] >> 6) << 4); }
}

void orc_dequantize_q4k(const uint8_t* b, float* y, int64_t nblocks) {
    for (int64_t i = 0; i < nblocks; ++i, b += Q4K_BYTES) {
        uint16_t dh, mh;
        memcpy(&dh, b, 2);
        memcpy(&mh, b + 2, 2);
        const float d = f16_to_f32(dh), dmin = f16_to_f32(mh);
        const uint8_t* q = b + 16;
        int is = 0;
        for (int j = 0; j < QK_K; j += 64) {
            uint8_t sc, m;
            scale_min_k4(is + 0, b + 4, &sc, &m);
            const float d1 = d * sc, m1 = dmin * m;
            scale_min_k4(is + 1, b + 4, &sc, &m);
            const float d2 = d * sc, m2 = dmin * m;
            for (int l = 0; l < 32; ++l) *y++ = d1 * (q[l] & 0xF) - m1;
            for (int l = 0; l < 32; ++l) *y++ = d2 * (q[l] >> 4) - m2;
            q += 32;
            is += 2;
        }
    }
}

void orc_dequantize_q6k(const uint8_t* b, float* y, int64_t nblocks) {
    for (int64_t i = 0; i < nblocks; ++i, b += Q6K_BYTES) {
        const uint8_t *ql = b, *qh = b + 128;
        const int8_t* sc = (const int8_t*)(b + 192);
        uint16_t dh;
        memcpy(&dh, b + 208, 2);
        const float d = f16_to_f32(dh);
        for (int n = 0; n < QK_K; n += 128) {
            for (int l = 0; l < 32; ++l) {
                const int is = l / 16;
                const int q1 = (int)((ql[l] & 0xF) | (((qh[l] >> 0) & 3) << 4)) - 32;
                const int q2 = (int)((ql[l + 32] & 0xF) | (((qh[l] >> 2) & 3) << 4)) - 32;
                const int q3 = (int)((ql[l] >> 4) | (((qh[l] >> 4) & 3) << 4)) - 32;
                const int q4 = (int)((ql[l + 32] >> 4) | (((qh[l] >> 6) & 3) << 4)) - 32;
                y[l] = d * sc[is] * q1;
                y[l + 32] = d * sc[is + 2] * q2;
                y[l + 64] = d * sc[is + 4] * q3;
                y[l + 96] = d * sc[is + 6] * q4;
            }
            y += 128; ql += 64; qh += 32; sc += 8;
        }
    }
}

/* Q8_K: f32 d; i8 qs[256]; i16 bsums[16]  -- ggml quantize_row_q8_K [EXT] */
void orc_quantize_q8k(const float* x, int k, float* d, int8_t* q, int16_t* bsums) {
    const int nb = k / QK_K;
    for (int i = 0; i < nb; ++i) {
        float max = 0, amax = 0;
        for (int j = 0; j < QK_K; ++j) {
            const float ax = fabsf(x[j]);
            if (ax > amax) { amax = ax; max = x[j]; }
        }
        if (amax == 0) {
            d[i] = 0;
            memset(q, 0, QK_K);
            memset(bsums, 0, 32);
        } else {
            const float iscale = -128.f / max;
            for (int j = 0; j < QK_K; ++j) {
                int v = (int)nearbyintf(iscale * x[j]);
                q[j] = (int8_t)(v > 127 ? 127 : v);
            }
            for (int j = 0; j < 16; ++j) {
                int s = 0;
                for (int l = 0; l < 16; ++l) s += q[j * 16 + l];
                bsums[j] = (int16_t)s;
            }
            d[i] = 1.f / iscale;
        }
        x += QK_K; q += QK_K; bsums += 16;
    }
}

/* The integer dot products below exist twice: a scalar statement (the definition) and an AVX2 statement of the SAME integer
 * arithmetic (candle's CPU backend runs ggml-style AVX2 kernels for these, so a scalar-only port would understate the CPU
 * baseline by an order of magnitude).  Integer sums are exact either way and the per-block float operations are identical,
 * so both give bit-identical results; tests/test_cpu_oracle.py checks that. */
#if defined(__AVX2__)
#include <immintrin.h>
static inline int32_t hsum_epi32(__m256i v) {
    __m128i s = _mm_add_epi32(_mm256_castsi256_si128(v), _mm256_extracti128_si256(v, 1));
    s = _mm_add_epi32(s, _mm_shuffle_epi32(s, 0x4E));
    s = _mm_add_epi32(s, _mm_shuffle_epi32(s, 0xB1));
    return _mm_cvtsi128_si32(s);
}
#endif

float orc_vec_dot_q4k_q8k_scalar(const uint8_t* w, int nb, const float* xd, const int8_t* xq, const int16_t* xb);
float orc_vec_dot_q6k_q8k_scalar(const uint8_t* w, int nb, const float* xd, const int8_t* xq);

/* A third statement of the same integer arithmetic for hosts with AVX-512 VNNI (the MI355X boxes' EPYC 9575F has it): the
 * u8 x s8 products go through vpdpbusd (four products per s32 lane, exact) instead of vpmaddubsw + vpmaddwd.  Chosen at run
 * time (`orc_isa`), bit-identical to the scalar definition like the AVX2 one (tests/test_cpu_oracle.py). */
#if defined(__x86_64__) && defined(__GNUC__)
#define ORC_HAVE_VNNI 1
#include <immintrin.h>
#define ORC_VNNI __attribute__((target("avx2,avx512f,avx512bw,avx512vl,avx512vnni")))
static int g_force_isa = -1;                       /* -1 auto, 0 scalar/AVX2 build default, 1 VNNI (tests) */
static int orc_use_vnni(void) {
    static int have = -1;
    if (have < 0) have = __builtin_cpu_supports("avx512vnni") && __builtin_cpu_supports("avx512vl") && __builtin_cpu_supports("avx512bw");
    if (g_force_isa == 0) return 0;
    return have;
}
void orc_force_isa(int v) { g_force_isa = v; }
const char* orc_isa(void) {
    if (orc_use_vnni()) return "avx512-vnni (vpdpbusd, 256-bit)";
#if defined(__AVX2__)
    return "avx2";
#else
    return "scalar";
#endif
}
ORC_VNNI static inline int32_t hsum_epi32_v(__m256i v) {
    __m128i s = _mm_add_epi32(_mm256_castsi256_si128(v), _mm256_extracti128_si256(v, 1));
    s = _mm_add_epi32(s, _mm_shuffle_epi32(s, 0x4E));
    s = _mm_add_epi32(s, _mm_shuffle_epi32(s, 0xB1));
    return _mm_cvtsi128_si32(s);
}
ORC_VNNI static float vec_dot_q4k_q8k_vnni(const uint8_t* w, int nb, const float* xd, const int8_t* xq, const int16_t* xb) {
    const __m256i m4 = _mm256_set1_epi8(0xF);
    float sumf = 0;
    for (int i = 0; i < nb; ++i, w += Q4K_BYTES, xq += QK_K, xb += 16) {
        uint16_t dh, mh;
        memcpy(&dh, w, 2);
        memcpy(&mh, w + 2, 2);
        const float d = f16_to_f32(dh) * xd[i], dmin = f16_to_f32(mh) * xd[i];
        const uint8_t* q4 = w + 16;
        __m256i acc = _mm256_setzero_si256();
        int32_t summ = 0;
        for (int p = 0; p < 4; ++p) {
            uint8_t sc0, m0, sc1, m1;
            scale_min_k4(2 * p, w + 4, &sc0, &m0);
            scale_min_k4(2 * p + 1, w + 4, &sc1, &m1);
            const __m256i q = _mm256_loadu_si256((const __m256i*)(q4 + 32 * p));
            const __m256i lo = _mm256_and_si256(q, m4), hi = _mm256_and_si256(_mm256_srli_epi16(q, 4), m4);
            const __m256i a = _mm256_loadu_si256((const __m256i*)(xq + 64 * p));
            const __m256i b = _mm256_loadu_si256((const __m256i*)(xq + 64 * p + 32));
            const __m256i z = _mm256_setzero_si256();
            acc = _mm256_add_epi32(acc, _mm256_mullo_epi32(_mm256_dpbusd_epi32(z, lo, a), _mm256_set1_epi32(sc0)));
            acc = _mm256_add_epi32(acc, _mm256_mullo_epi32(_mm256_dpbusd_epi32(z, hi, b), _mm256_set1_epi32(sc1)));
            summ += (xb[4 * p] + xb[4 * p + 1]) * m0 + (xb[4 * p + 2] + xb[4 * p + 3]) * m1;
        }
        sumf += d * (float)hsum_epi32_v(acc) - dmin * (float)summ;
    }
    return sumf;
}
ORC_VNNI static float vec_dot_q6k_q8k_vnni(const uint8_t* w, int nb, const float* xd, const int8_t* xq) {
    const __m256i m4 = _mm256_set1_epi8(0xF), m2 = _mm256_set1_epi8(3), m32 = _mm256_set1_epi8(32);
    float sumf = 0;
    for (int i = 0; i < nb; ++i, w += Q6K_BYTES, xq += QK_K) {
        const uint8_t *ql = w, *qh = w + 128;
        const int8_t* sc = (const int8_t*)(w + 192);
        uint16_t dh;
        memcpy(&dh, w + 208, 2);
        const float d = f16_to_f32(dh) * xd[i];
        __m256i acc = _mm256_setzero_si256();
        const int8_t* q8 = xq;
        for (int n = 0; n < 2; ++n) {
            const __m256i la = _mm256_loadu_si256((const __m256i*)ql), lb = _mm256_loadu_si256((const __m256i*)(ql + 32));
            const __m256i h = _mm256_loadu_si256((const __m256i*)qh);
            __m256i q[4];
            q[0] = _mm256_or_si256(_mm256_and_si256(la, m4), _mm256_slli_epi16(_mm256_and_si256(h, m2), 4));
            q[1] = _mm256_or_si256(_mm256_and_si256(lb, m4), _mm256_slli_epi16(_mm256_and_si256(_mm256_srli_epi16(h, 2), m2), 4));
            q[2] = _mm256_or_si256(_mm256_and_si256(_mm256_srli_epi16(la, 4), m4), _mm256_slli_epi16(_mm256_and_si256(_mm256_srli_epi16(h, 4), m2), 4));
            q[3] = _mm256_or_si256(_mm256_and_si256(_mm256_srli_epi16(lb, 4), m4), _mm256_slli_epi16(_mm256_and_si256(_mm256_srli_epi16(h, 6), m2), 4));
            for (int g = 0; g < 4; ++g) {
                const __m256i a = _mm256_loadu_si256((const __m256i*)(q8 + 32 * g));
                const __m256i z = _mm256_setzero_si256();
                /* sum (q - 32) a = sum q a - 32 sum a, four products per s32 lane; lanes 0..3 belong to scale 2g, 4..7 to 2g+1 */
                const __m256i p32 = _mm256_sub_epi32(_mm256_dpbusd_epi32(z, q[g], a), _mm256_dpbusd_epi32(z, m32, a));
                const __m256i scv = _mm256_set_m128i(_mm_set1_epi32(sc[2 * g + 1]), _mm_set1_epi32(sc[2 * g]));
                acc = _mm256_add_epi32(acc, _mm256_mullo_epi32(p32, scv));
            }
            ql += 64; qh += 32; sc += 8; q8 += 128;
        }
        sumf += d * (float)hsum_epi32_v(acc);
    }
    return sumf;
}
#else
const char* orc_isa(void) { return "scalar"; }
void orc_force_isa(int v) { (void)v; }
#endif

float orc_vec_dot_q4k_q8k(const uint8_t* w, int nb, const float* xd, const int8_t* xq, const int16_t* xb) {
#ifdef ORC_HAVE_VNNI
    if (orc_use_vnni()) return vec_dot_q4k_q8k_vnni(w, nb, xd, xq, xb);
#endif
#if defined(__AVX2__)
    const __m256i m4 = _mm256_set1_epi8(0xF);
    float sumf = 0;
    for (int i = 0; i < nb; ++i, w += Q4K_BYTES, xq += QK_K, xb += 16) {
        uint16_t dh, mh;
        memcpy(&dh, w, 2);
        memcpy(&mh, w + 2, 2);
        const float d = f16_to_f32(dh) * xd[i], dmin = f16_to_f32(mh) * xd[i];
        const uint8_t* q4 = w + 16;
        __m256i acc = _mm256_setzero_si256();
        int32_t summ = 0;
        for (int p = 0; p < 4; ++p) {                    /* 64 weights: low nibbles = sub-block 2p, high = 2p+1 */
            uint8_t sc0, m0, sc1, m1;
            scale_min_k4(2 * p, w + 4, &sc0, &m0);
            scale_min_k4(2 * p + 1, w + 4, &sc1, &m1);
            const __m256i q = _mm256_loadu_si256((const __m256i*)(q4 + 32 * p));
            const __m256i lo = _mm256_and_si256(q, m4), hi = _mm256_and_si256(_mm256_srli_epi16(q, 4), m4);
            const __m256i a = _mm256_loadu_si256((const __m256i*)(xq + 64 * p));
            const __m256i b = _mm256_loadu_si256((const __m256i*)(xq + 64 * p + 32));
            /* u8 x s8 pair sums fit s16 (<= 2*15*127); times the 6-bit scale in s32 lanes */
            acc = _mm256_add_epi32(acc, _mm256_madd_epi16(_mm256_maddubs_epi16(lo, a), _mm256_set1_epi16(sc0)));
            acc = _mm256_add_epi32(acc, _mm256_madd_epi16(_mm256_maddubs_epi16(hi, b), _mm256_set1_epi16(sc1)));
            summ += (xb[4 * p] + xb[4 * p + 1]) * m0 + (xb[4 * p + 2] + xb[4 * p + 3]) * m1;
        }
        sumf += d * (float)hsum_epi32(acc) - dmin * (float)summ;
    }
    return sumf;
#else
    return orc_vec_dot_q4k_q8k_scalar(w, nb, xd, xq, xb);
#endif
}

float orc_vec_dot_q4k_q8k_scalar(const uint8_t* w, int nb, const float* xd, const int8_t* xq, const int16_t* xb) {
    float sumf = 0;
    for (int i = 0; i < nb; ++i, w += Q4K_BYTES, xq += QK_K, xb += 16) {
        uint16_t dh, mh;
        memcpy(&dh, w, 2);
        memcpy(&mh, w + 2, 2);
        const float d = f16_to_f32(dh) * xd[i], dmin = f16_to_f32(mh) * xd[i];
        const uint8_t* q4 = w + 16;
        int32_t sumi = 0, summ = 0;
        for (int j = 0; j < 8; ++j) {
            uint8_t sc, m;
            scale_min_k4(j, w + 4, &sc, &m);
            const uint8_t* qs = q4 + 32 * (j / 2);
            const int8_t* q8 = xq + 32 * j;
            int32_t s = 0;
            if (j & 1) for (int l = 0; l < 32; ++l) s += (qs[l] >> 4) * q8[l];
            else for (int l = 0; l < 32; ++l) s += (qs[l] & 0xF) * q8[l];
            sumi += s * sc;
            summ += (xb[2 * j] + xb[2 * j + 1]) * m;
        }
        sumf += d * (float)sumi - dmin * (float)summ;
    }
    return sumf;
}

float orc_vec_dot_q6k_q8k(const uint8_t* w, int nb, const float* xd, const int8_t* xq) {
#ifdef ORC_HAVE_VNNI
    if (orc_use_vnni()) return vec_dot_q6k_q8k_vnni(w, nb, xd, xq);
#endif
#if defined(__AVX2__)
    const __m256i m4 = _mm256_set1_epi8(0xF), m2 = _mm256_set1_epi8(3), m32 = _mm256_set1_epi8(32);
    float sumf = 0;
    for (int i = 0; i < nb; ++i, w += Q6K_BYTES, xq += QK_K) {
        const uint8_t *ql = w, *qh = w + 128;
        const int8_t* sc = (const int8_t*)(w + 192);
        uint16_t dh;
        memcpy(&dh, w + 208, 2);
        const float d = f16_to_f32(dh) * xd[i];
        __m256i acc = _mm256_setzero_si256();
        const int8_t* q8 = xq;
        for (int n = 0; n < 2; ++n) {
            const __m256i la = _mm256_loadu_si256((const __m256i*)ql), lb = _mm256_loadu_si256((const __m256i*)(ql + 32));
            const __m256i h = _mm256_loadu_si256((const __m256i*)qh);
            /* the four 32-weight groups of this half, as unsigned 6-bit codes (the -32 is taken off below) */
            __m256i q[4];
            q[0] = _mm256_or_si256(_mm256_and_si256(la, m4), _mm256_slli_epi16(_mm256_and_si256(h, m2), 4));
            q[1] = _mm256_or_si256(_mm256_and_si256(lb, m4), _mm256_slli_epi16(_mm256_and_si256(_mm256_srli_epi16(h, 2), m2), 4));
            q[2] = _mm256_or_si256(_mm256_and_si256(_mm256_srli_epi16(la, 4), m4), _mm256_slli_epi16(_mm256_and_si256(_mm256_srli_epi16(h, 4), m2), 4));
            q[3] = _mm256_or_si256(_mm256_and_si256(_mm256_srli_epi16(lb, 4), m4), _mm256_slli_epi16(_mm256_and_si256(_mm256_srli_epi16(h, 6), m2), 4));
            for (int g = 0; g < 4; ++g) {
                const __m256i a = _mm256_loadu_si256((const __m256i*)(q8 + 32 * g));
                /* sum (q - 32) * a  =  sum q*a - 32 * sum a, pairwise in s16 (|.| <= 2*63*127 + 2*32*127 < 32768) */
                const __m256i p16 = _mm256_sub_epi16(_mm256_maddubs_epi16(q[g], a), _mm256_maddubs_epi16(m32, a));
                /* scale index: 16 weights per scale -> s16 lanes 0..7 use sc[2g], lanes 8..15 use sc[2g+1] */
                const __m256i scv = _mm256_set_m128i(_mm_set1_epi16(sc[2 * g + 1]), _mm_set1_epi16(sc[2 * g]));
                acc = _mm256_add_epi32(acc, _mm256_madd_epi16(p16, scv));
            }
            ql += 64; qh += 32; sc += 8; q8 += 128;
        }
        sumf += d * (float)hsum_epi32(acc);
    }
    return sumf;
#else
    return orc_vec_dot_q6k_q8k_scalar(w, nb, xd, xq);
#endif
}

float orc_vec_dot_q6k_q8k_scalar(const uint8_t* w, int nb, const float* xd, const int8_t* xq) {
    float sumf = 0;
    for (int i = 0; i < nb; ++i, w += Q6K_BYTES, xq += QK_K) {
        const uint8_t *ql = w, *qh = w + 128;
        const int8_t* sc = (const int8_t*)(w + 192);
        uint16_t dh;
        memcpy(&dh, w + 208, 2);
        const float d = f16_to_f32(dh) * xd[i];
        int32_t sumi = 0;
        const int8_t* q8 = xq;
        for (int n = 0; n < 2; ++n) {
            int32_t s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int l = 0; l < 32; ++l) {
                const int is = l / 16;
                s[is] += ((int)((ql[l] & 0xF) | (((qh[l] >> 0) & 3) << 4)) - 32) * q8[l];
                s[is + 2] += ((int)((ql[l + 32] & 0xF) | (((qh[l] >> 2) & 3) << 4)) - 32) * q8[l + 32];
                s[is + 4] += ((int)((ql[l] >> 4) | (((qh[l] >> 4) & 3) << 4)) - 32) * q8[l + 64];
                s[is + 6] += ((int)((ql[l + 32] >> 4) | (((qh[l] >> 6) & 3) << 4)) - 32) * q8[l + 96];
            }
            for (int j = 0; j < 8; ++j) sumi += s[j] * sc[j];
            ql += 64; qh += 32; sc += 8; q8 += 128;
        }
        sumf += d * (float)sumi;
    }
    return sumf;
}

/* y[T,N] = x[T,K] . dequant(W[N,K])^T.  o2 != 0: candle-CPU-faithful (Q8_K activations, integer dot);
 * o2 == 0: dequantise each row, accumulate in double (oracle O1). */
static void qmatmul_o1f(const uint8_t* w, int type, int N, int K, const float* x, int T, float* y);
void orc_qmatmul(const uint8_t* w, int type, int N, int K, const float* x, int T, float* y, int o2) {
    const int nb = K / QK_K, bb = type == T_Q4K ? Q4K_BYTES : Q6K_BYTES;
    if (o2 == 2) { qmatmul_o1f(w, type, N, K, x, T, y); return; }
    if (o2) {
        float* xd = (float*)malloc((size_t)T * nb * 4);
        int8_t* xq = (int8_t*)malloc((size_t)T * K);
        int16_t* xb = (int16_t*)malloc((size_t)T * nb * 16 * 2);
        for (int t = 0; t < T; ++t)
            orc_quantize_q8k(x + (size_t)t * K, K, xd + (size_t)t * nb, xq + (size_t)t * K, xb + (size_t)t * nb * 16);
#pragma omp parallel for schedule(static)
        for (int n = 0; n < N; ++n) {
            const uint8_t* row = w + (size_t)n * nb * bb;
            for (int t = 0; t < T; ++t)
                y[(size_t)t * N + n] = type == T_Q4K
                    ? orc_vec_dot_q4k_q8k(row, nb, xd + (size_t)t * nb, xq + (size_t)t * K, xb + (size_t)t * nb * 16)
                    : orc_vec_dot_q6k_q8k(row, nb, xd + (size_t)t * nb, xq + (size_t)t * K);
        }
        free(xd); free(xq); free(xb);
    } else {
#pragma omp parallel
        {
            float* wr = (float*)malloc((size_t)K * 4);
#pragma omp for schedule(static)
            for (int n = 0; n < N; ++n) {
                const uint8_t* row = w + (size_t)n * nb * bb;
                if (type == T_Q4K) orc_dequantize_q4k(row, wr, nb); else orc_dequantize_q6k(row, wr, nb);
                for (int t = 0; t < T; ++t) {
                    double acc = 0;
                    const float* xr = x + (size_t)t * K;
                    for (int k = 0; k < K; ++k) acc += (double)xr[k] * (double)wr[k];
                    y[(size_t)t * N + n] = (float)acc;
                }
            }
            free(wr);
        }
    }
}

/* O1f: the O1 definition (dequantise the row, dot with the f32 activations) with f32 accumulation in 8 (AVX2) or 1
 * partial sums instead of one double, blocked 16 rows x all tokens so a many-token product (the full-size parity
 * leg's batch-32 step and 2048-token prompt step, tests/fullsize_parity.py) finishes in seconds.  Differs from O1 by
 * f32 summation noise only (~1e-6 relative, tests/test_cpu_oracle.py); same call sites as O1. */
#define O1F_RB 16
static void qmatmul_o1f(const uint8_t* w, int type, int N, int K, const float* x, int T, float* y) {
    const int nb = K / QK_K, bb = type == T_Q4K ? Q4K_BYTES : Q6K_BYTES;
#pragma omp parallel
    {
        float* wr = (float*)aligned_alloc(64, (size_t)O1F_RB * K * 4);
#pragma omp for schedule(dynamic, 1)
        for (int n0 = 0; n0 < N; n0 += O1F_RB) {
            const int nr = N - n0 < O1F_RB ? N - n0 : O1F_RB;
            for (int r = 0; r < nr; ++r) {
                const uint8_t* row = w + (size_t)(n0 + r) * nb * bb;
                if (type == T_Q4K) orc_dequantize_q4k(row, wr + (size_t)r * K, nb);
                else orc_dequantize_q6k(row, wr + (size_t)r * K, nb);
            }
            for (int t = 0; t < T; ++t) {
                const float* xr = x + (size_t)t * K;
                int r = 0;
#if defined(__AVX2__) && defined(__FMA__)
                for (; r + 4 <= nr; r += 4) {
                    const float *w0 = wr + (size_t)r * K, *w1 = w0 + K, *w2 = w1 + K, *w3 = w2 + K;
                    __m256 a0 = _mm256_setzero_ps(), a1 = a0, a2 = a0, a3 = a0;
                    for (int k = 0; k < K; k += 8) {
                        const __m256 xv = _mm256_loadu_ps(xr + k);
                        a0 = _mm256_fmadd_ps(xv, _mm256_load_ps(w0 + k), a0);
                        a1 = _mm256_fmadd_ps(xv, _mm256_load_ps(w1 + k), a1);
                        a2 = _mm256_fmadd_ps(xv, _mm256_load_ps(w2 + k), a2);
                        a3 = _mm256_fmadd_ps(xv, _mm256_load_ps(w3 + k), a3);
                    }
                    __m256 acc[4] = {a0, a1, a2, a3};
                    for (int j = 0; j < 4; ++j) {
                        __m128 s = _mm_add_ps(_mm256_castps256_ps128(acc[j]), _mm256_extractf128_ps(acc[j], 1));
                        s = _mm_add_ps(s, _mm_movehl_ps(s, s));
                        s = _mm_add_ss(s, _mm_shuffle_ps(s, s, 1));
                        y[(size_t)t * N + n0 + r + j] = _mm_cvtss_f32(s);
                    }
                }
#endif
                for (; r < nr; ++r) {
                    const float* wv = wr + (size_t)r * K;
                    float acc = 0;
                    for (int k = 0; k < K; ++k) acc += xr[k] * wv[k];
                    y[(size_t)t * N + n0 + r] = acc;
                }
            }
        }
        free(wr);
    }
}

/* ---- small ops --------------------------------------------------------------------------------- */
void orc_rms_norm(const float* x, const float* w, float eps, int T, int hid, float* y) {
    for (int t = 0; t < T; ++t) {
        double ss = 0;
        for (int i = 0; i < hid; ++i) ss += (double)x[(size_t)t * hid + i] * x[(size_t)t * hid + i];
        const double inv = 1.0 / sqrt(ss / hid + eps);
        for (int i = 0; i < hid; ++i) y[(size_t)t * hid + i] = (float)(x[(size_t)t * hid + i] * inv * w[i]);
    }
}

/* interleaved rope (candle rope_i) in place; x [T, H, D]; cos/sin [max_seq, D/2] */
void orc_rope_i(float* x, const float* cosT, const float* sinT, const int64_t* pos, int T, int H, int D) {
    for (int t = 0; t < T; ++t)
        for (int h = 0; h < H; ++h) {
            float* v = x + ((size_t)t * H + h) * D;
            for (int i = 0; i < D / 2; ++i) {
                const float c = cosT[pos[t] * (D / 2) + i], s = sinT[pos[t] * (D / 2) + i];
                const float x0 = v[2 * i], x1 = v[2 * i + 1];
                v[2 * i] = x0 * c - x1 * s;
                v[2 * i + 1] = x0 * s + x1 * c;
            }
        }
}

/* ---- the model ---------------------------------------------------------------------------------- */
typedef struct {
    int32_t hidden, n_layers, n_heads, n_kv_heads, head_dim, intermediate, vocab, max_seq, block_size;
    float rms_eps, rope_theta;
} orc_cfg;

typedef struct { const uint8_t* w; int type, n, k; } orc_qw;
typedef struct {
    orc_cfg c;
    orc_qw* lw;            /* [n_layers][7] */
    float** norms;         /* [n_layers][2] */
    const float* tok_embd; /* f32 [vocab, hidden] or NULL (then the embedding row is synthesised) */
    const float* out_norm;
    orc_qw out;
    float *cosT, *sinT;
    uint8_t* owned;        /* random weights allocated by orc_llama_create_random */
    float* owned_f;
    float* trace;          /* optional: the residual stream at every layer entry + after the last layer, [n_layers+1][B][hidden] */
    float *tr_q, *tr_att, *tr_mid, *tr_h;   /* optional per-layer intermediates: [L][B][H*D] x2, [L][B][hidden], [L][B][I] */
} orc_model;

void* orc_llama_create(const orc_cfg* c) {
    orc_model* m = (orc_model*)calloc(1, sizeof(orc_model));
    m->c = *c;
    m->lw = (orc_qw*)calloc((size_t)c->n_layers * 7, sizeof(orc_qw));
    m->norms = (float**)calloc((size_t)c->n_layers * 2, sizeof(float*));
    const int half = c->head_dim / 2;
    m->cosT = (float*)malloc((size_t)c->max_seq * half * 4);
    m->sinT = (float*)malloc((size_t)c->max_seq * half * 4);
    for (int i = 0; i < half; ++i) {
        /* calculate_default_inv_freq (rotary_emb.rs:14-19): base^(i/dim) in f64, cast to f32, reciprocal taken IN f32.  (A f64
         * reciprocal differs by one f32 ulp for some i; at position 4096 that is half a milliradian on the fastest pairs --
         * found by the full-size parity leg, tests/test_gpu_fullsize.py.) */
        const float inv = 1.0f / (float)pow((double)c->rope_theta, (double)(2 * i) / (double)c->head_dim);
        for (int p = 0; p < c->max_seq; ++p) {
            const float th = (float)p * inv;
            m->cosT[(size_t)p * half + i] = (float)cos((double)th);
            m->sinT[(size_t)p * half + i] = (float)sin((double)th);
        }
    }
    return m;
}
void orc_llama_destroy(void* mp) {
    orc_model* m = (orc_model*)mp;
    if (!m) return;
    free(m->lw); free(m->norms); free(m->cosT); free(m->sinT); free(m->owned); free(m->owned_f); free(m);
}
/* decode steps record `layer_in` of every layer (quantized_llama.rs:438) and the final hidden state into `trace`
 * ([n_layers+1][B][hidden], borrowed; NULL = off) so a test can compare the GPU path one layer at a time */
void orc_llama_set_trace(void* mp, float* trace) { ((orc_model*)mp)->trace = trace; }
/* ... and the values at the rounding points inside every layer: q after RoPE rounded to bf16 (attention.rs:977-981), the
 * attention output rounded to bf16 (:995-1003), the stream after the attention residual (quantized_llama.rs:464) and the
 * MLP intermediate silu(w1 x) * w3 x (:33-36) -- so each launch group can be checked from the oracle's own inputs */
void orc_llama_set_trace_parts(void* mp, float* q, float* att, float* mid, float* h) {
    orc_model* m = (orc_model*)mp;
    m->tr_q = q; m->tr_att = att; m->tr_mid = mid; m->tr_h = h;
}
/* which: 0 wq 1 wk 2 wv 3 wo 4 w1 5 w2 6 w3 ; layer -1, which 11: output.  Pointers are borrowed. */
void orc_llama_set_qweight(void* mp, int layer, int which, int type, const uint8_t* blocks, int n, int k) {
    orc_model* m = (orc_model*)mp;
    orc_qw q = {blocks, type, n, k};
    if (layer < 0) m->out = q; else m->lw[(size_t)layer * 7 + which] = q;
}
/* which: 7 attn_norm 8 ffn_norm ; layer -1: 9 tok_embd 10 output_norm */
void orc_llama_set_f32(void* mp, int layer, int which, const float* p) {
    orc_model* m = (orc_model*)mp;
    if (layer < 0) { if (which == 9) m->tok_embd = p; else m->out_norm = p; }
    else m->norms[(size_t)layer * 2 + (which == 7 ? 0 : 1)] = (float*)p;
}

static uint64_t xs64(uint64_t* s) { uint64_t x = *s; x ^= x << 13; x ^= x >> 7; x ^= x << 17; return *s = x; }

/* Random VALID weights (Q4_K_M-style type mixture decided by the caller through types[]) for the CPU
 * baseline timing: bytes from xorshift, f16 super-block scales set to sane constants. */
/* attention arithmetic of orc_llama_decode: 0 = f32 scores / softmax / P.V, output rounded to bf16 (the oracle's parity
 * target); 1 = what the reference's CPU path literally does with its bf16 q, k, v tensors (models/mod.rs:1288-1306): the
 * score matmul returns bf16, `* scale` rounds again, softmax_last_dim writes bf16 probabilities, the P.V matmul returns
 * bf16 [EXT candle CPU matmul / softmax: f32 inside, dtype of the tensor outside].  Used to measure how far the
 * reference's own rounding points move the logits (tests/fullsize_parity.py). */
static int g_attn_bf16 = 0;
static int g_attn_fast = 0;      /* 1: vectorised f32 QK dot (timed cpu_baseline only; never the parity oracle) */
void orc_set_attn_fast(int on) { g_attn_fast = on ? 1 : 0; }
void orc_llama_set_attn_bf16(int on) { g_attn_bf16 = on; }
static float g_fill_scale = 1.0f;   /* orc_llama_set_fill_scale: multiplies the super-block scales of the NEXT fill */
void orc_llama_set_fill_scale(float s) { g_fill_scale = s > 0 ? s : 1.0f; }
static uint16_t f32_to_f16(float f) {   /* round to nearest even, subnormals included (values here are tiny positives) */
    uint32_t u;
    memcpy(&u, &f, 4);
    const uint32_t sign = (u >> 16) & 0x8000u;
    int32_t e = (int32_t)((u >> 23) & 0xFF) - 127 + 15;
    uint32_t m = u & 0x7FFFFFu;
    if (e >= 31) return (uint16_t)(sign | 0x7C00u);
    if (e <= 0) {
        if (e < -10) return (uint16_t)sign;
        m |= 0x800000u;
        const int sh = 14 - e;
        uint32_t r = m >> sh;
        const uint32_t rem = m & ((1u << sh) - 1), half = 1u << (sh - 1);
        if (rem > half || (rem == half && (r & 1))) ++r;
        return (uint16_t)(sign | r);
    }
    uint32_t r = ((uint32_t)e << 10) | (m >> 13);
    const uint32_t rem = m & 0x1FFFu;
    if (rem > 0x1000u || (rem == 0x1000u && (r & 1))) ++r;
    return (uint16_t)(sign | r);
}
static void fill_random(uint8_t* p, int type, int64_t nblocks, uint64_t* seed) {
    const int bb = type == T_Q4K ? Q4K_BYTES : Q6K_BYTES;
    uint64_t* q = (uint64_t*)p;
    const int64_t n8 = nblocks * bb / 8;
    for (int64_t i = 0; i < n8; ++i) q[i] = xs64(seed);
    for (int64_t i = n8 * 8; i < nblocks * bb; ++i) p[i] = (uint8_t)xs64(seed);
    uint16_t d4 = 0x0A8E /* 2e-4 */, dm4 = 0x1625 /* 1.5e-3 */, d6 = 0x00D2 /* 1.25e-5 */;
    if (g_fill_scale != 1.0f) { d4 = f32_to_f16(2e-4f * g_fill_scale); dm4 = f32_to_f16(1.5e-3f * g_fill_scale); d6 = f32_to_f16(1.25e-5f * g_fill_scale); }
    for (int64_t i = 0; i < nblocks; ++i) {
        if (type == T_Q4K) { memcpy(p + i * bb, &d4, 2); memcpy(p + i * bb + 2, &dm4, 2); }
        else memcpy(p + i * bb + 208, &d6, 2);
    }
}

/* allocate + fill all weights; types: int[n_layers*7 + 1] (last = output) */
int orc_llama_fill_random(void* mp, const int32_t* types, uint64_t seed) {
    orc_model* m = (orc_model*)mp;
    const orc_cfg* c = &m->c;
    const int H = c->n_heads, Hkv = c->n_kv_heads, D = c->head_dim, hid = c->hidden, I = c->intermediate;
    const int rows[7] = {H * D, Hkv * D, Hkv * D, hid, I, hid, I};
    const int cols[7] = {hid, hid, hid, H * D, hid, I, hid};
    size_t total = 0;
    for (int l = 0; l < c->n_layers; ++l)
        for (int w = 0; w < 7; ++w)
            total += (size_t)rows[w] * (cols[w] / QK_K) * (types[l * 7 + w] == T_Q4K ? Q4K_BYTES : Q6K_BYTES);
    const int ot = types[c->n_layers * 7];
    total += (size_t)c->vocab * (hid / QK_K) * (ot == T_Q4K ? Q4K_BYTES : Q6K_BYTES);
    m->owned = (uint8_t*)malloc(total);
    m->owned_f = (float*)malloc(((size_t)c->n_layers * 2 + 1) * hid * 4);
    if (!m->owned || !m->owned_f) return -1;
    uint64_t s = seed ? seed : 88172645463325252ull;
    uint8_t* p = m->owned;
    for (int l = 0; l < c->n_layers; ++l) {
        for (int w = 0; w < 7; ++w) {
            const int t = types[l * 7 + w];
            const int64_t nbk = (int64_t)rows[w] * (cols[w] / QK_K);
            fill_random(p, t, nbk, &s);
            orc_llama_set_qweight(m, l, w, t, p, rows[w], cols[w]);
            p += nbk * (t == T_Q4K ? Q4K_BYTES : Q6K_BYTES);
        }
        for (int k = 0; k < 2; ++k) {
            float* nw = m->owned_f + ((size_t)l * 2 + k) * hid;
            for (int i = 0; i < hid; ++i) nw[i] = 1.0f;
            m->norms[(size_t)l * 2 + k] = nw;
        }
    }
    fill_random(p, ot, (int64_t)c->vocab * (hid / QK_K), &s);
    orc_llama_set_qweight(m, -1, 11, ot, p, c->vocab, hid);
    float* on = m->owned_f + (size_t)c->n_layers * 2 * hid;
    for (int i = 0; i < hid; ++i) on[i] = 1.0f;
    m->out_norm = on;
    m->tok_embd = NULL;
    return 0;
}

/* where a decode step spends its wall time (seconds, accumulated over calls): [0] quantised mat-vecs, [1] attention, [2] the rest
 * (norms, RoPE, cache write, residuals, SiLU) -- bench.py's cpu_baseline reports them (is the host baseline bound by a
 * serial section or by a CPU quota?) */
static double g_phase[3];
static double orc_now(void) { return omp_get_wtime(); }
void orc_llama_phase_times(double* out3, int reset) {
    for (int i = 0; i < 3; ++i) { out3[i] = g_phase[i]; if (reset) g_phase[i] = 0; }
}
#define ORC_MM(...) do { const double t_ = orc_now(); orc_qmatmul(__VA_ARGS__); g_phase[0] += orc_now() - t_; } while (0)

/* One decode step (flash KV layout [NB, bs, Hkv, D], bf16 bit patterns).  logits f32 [B, vocab]. */
void orc_llama_decode(void* mp, const uint32_t* tokens, const int64_t* positions, const int64_t* slots,
                      const uint32_t* bt, const uint32_t* ctx, int B, int max_blocks, uint16_t** kcache,
                      uint16_t** vcache, float* logits, int o2) {
    orc_model* m = (orc_model*)mp;
    const orc_cfg* c = &m->c;
    const int H = c->n_heads, Hkv = c->n_kv_heads, D = c->head_dim, hid = c->hidden, I = c->intermediate;
    const int G = H / Hkv, bs = c->block_size;
    float* xs = (float*)malloc((size_t)B * hid * 4);
    float* xn = (float*)malloc((size_t)B * hid * 4);
    float* q = (float*)malloc((size_t)B * H * D * 4);
    float* k = (float*)malloc((size_t)B * Hkv * D * 4);
    float* v = (float*)malloc((size_t)B * Hkv * D * 4);
    float* att = (float*)malloc((size_t)B * H * D * 4);
    float* g = (float*)malloc((size_t)B * I * 4);
    float* u = (float*)malloc((size_t)B * I * 4);
    float* tmp = (float*)malloc((size_t)B * hid * 4);
    for (int b = 0; b < B; ++b)
        for (int i = 0; i < hid; ++i)
            xs[(size_t)b * hid + i] = m->tok_embd ? m->tok_embd[(size_t)tokens[b] * hid + i]
                                                  : 0.02f * sinf((float)(tokens[b] % 977) * 0.37f + (float)i * 0.011f);
    const float scale = 1.0f / sqrtf((float)D);
    const double t_step = orc_now(), mm0 = g_phase[0], at0 = g_phase[1];
    for (int l = 0; l < c->n_layers; ++l) {
        const orc_qw* W = m->lw + (size_t)l * 7;
        if (m->trace) memcpy(m->trace + (size_t)l * B * hid, xs, (size_t)B * hid * 4);
        orc_rms_norm(xs, m->norms[(size_t)l * 2], c->rms_eps, B, hid, xn);
        ORC_MM(W[0].w, W[0].type, W[0].n, W[0].k, xn, B, q, o2);
        ORC_MM(W[1].w, W[1].type, W[1].n, W[1].k, xn, B, k, o2);
        ORC_MM(W[2].w, W[2].type, W[2].n, W[2].k, xn, B, v, o2);
        orc_rope_i(q, m->cosT, m->sinT, positions, B, H, D);
        orc_rope_i(k, m->cosT, m->sinT, positions, B, Hkv, D);
        for (int b = 0; b < B; ++b) {                      /* cast to bf16 + cache write (attention.rs:977-995) */
            if (slots[b] < 0) continue;
            for (int i = 0; i < Hkv * D; ++i) {
                kcache[l][(size_t)slots[b] * Hkv * D + i] = f32_to_bf16(k[(size_t)b * Hkv * D + i]);
                vcache[l][(size_t)slots[b] * Hkv * D + i] = f32_to_bf16(v[(size_t)b * Hkv * D + i]);
            }
        }
        const double t_att = orc_now();
#pragma omp parallel for collapse(2) schedule(static)
        for (int b = 0; b < B; ++b)
            for (int h = 0; h < H; ++h) {
                const int n = (int)ctx[b], hk = h / G;
                float* sc = (float*)malloc((size_t)n * 4);
                float qb[512];
                for (int d = 0; d < D; ++d) qb[d] = round_bf16(q[((size_t)b * H + h) * D + d]);
                float mx = -1e30f;
                for (int t = 0; t < n; ++t) {
                    const size_t blk = bt[(size_t)b * max_blocks + t / bs];
                    const uint16_t* kr = kcache[l] + ((blk * bs + t % bs) * Hkv + hk) * D;
                    float s = 0;
#if defined(__AVX2__) && defined(__FMA__)
                    if (g_attn_fast && !g_attn_bf16 && (D & 7) == 0) {
                        /* the TIMED cpu_baseline only (orc_set_attn_fast(1), bench.py): eight partial sums -- a CPU backend's matmul does
                         * not walk a 128-element dot through one dependent FMA chain (that chain alone was 60 % of the baseline's step
                         * at ctx 4096).  The PARITY oracle (flag off, the default) keeps the scalar index order below on every host:
                         * its numerics must not depend on the ISA the checker happens to be compiled for. */
                        __m256 acc = _mm256_setzero_ps();
                        for (int d = 0; d < D; d += 8) {
                            const __m256i kb = _mm256_slli_epi32(_mm256_cvtepu16_epi32(_mm_loadu_si128((const __m128i*)(kr + d))), 16);
                            acc = _mm256_fmadd_ps(_mm256_loadu_ps(qb + d), _mm256_castsi256_ps(kb), acc);
                        }
                        __m128 r4 = _mm_add_ps(_mm256_castps256_ps128(acc), _mm256_extractf128_ps(acc, 1));
                        r4 = _mm_add_ps(r4, _mm_movehl_ps(r4, r4));
                        r4 = _mm_add_ss(r4, _mm_shuffle_ps(r4, r4, 1));
                        s = _mm_cvtss_f32(r4);
                    } else
#endif
                    for (int d = 0; d < D; ++d) s += qb[d] * bf16_to_f32(kr[d]);
                    sc[t] = g_attn_bf16 ? round_bf16(round_bf16(s) * scale) : s * scale;
                    if (sc[t] > mx) mx = sc[t];
                }
                double den = 0;
                for (int t = 0; t < n; ++t) { sc[t] = expf(sc[t] - mx); den += sc[t]; }
                float* o = att + ((size_t)b * H + h) * D;
                for (int d = 0; d < D; ++d) o[d] = 0;
                for (int t = 0; t < n; ++t) {
                    const size_t blk = bt[(size_t)b * max_blocks + t / bs];
                    const uint16_t* vr = vcache[l] + ((blk * bs + t % bs) * Hkv + hk) * D;
                    float p = (float)(sc[t] / den);
                    if (g_attn_bf16) p = round_bf16(p);
                    for (int d = 0; d < D; ++d) o[d] += p * bf16_to_f32(vr[d]);
                }
                for (int d = 0; d < D; ++d) o[d] = round_bf16(o[d]);
                free(sc);
            }
        g_phase[1] += orc_now() - t_att;
        if (m->tr_q) for (size_t i = 0; i < (size_t)B * H * D; ++i) m->tr_q[(size_t)l * B * H * D + i] = round_bf16(q[i]);
        if (m->tr_att) memcpy(m->tr_att + (size_t)l * B * H * D, att, (size_t)B * H * D * 4);
        ORC_MM(W[3].w, W[3].type, W[3].n, W[3].k, att, B, tmp, o2);
        for (size_t i = 0; i < (size_t)B * hid; ++i) xs[i] += tmp[i];
        if (m->tr_mid) memcpy(m->tr_mid + (size_t)l * B * hid, xs, (size_t)B * hid * 4);
        orc_rms_norm(xs, m->norms[(size_t)l * 2 + 1], c->rms_eps, B, hid, xn);
        ORC_MM(W[4].w, W[4].type, W[4].n, W[4].k, xn, B, g, o2);
        ORC_MM(W[6].w, W[6].type, W[6].n, W[6].k, xn, B, u, o2);
        for (size_t i = 0; i < (size_t)B * I; ++i) g[i] = g[i] / (1.f + expf(-g[i])) * u[i];
        if (m->tr_h) memcpy(m->tr_h + (size_t)l * B * I, g, (size_t)B * I * 4);
        ORC_MM(W[5].w, W[5].type, W[5].n, W[5].k, g, B, tmp, o2);
        for (size_t i = 0; i < (size_t)B * hid; ++i) xs[i] += tmp[i];
    }
    if (m->trace) memcpy(m->trace + (size_t)c->n_layers * B * hid, xs, (size_t)B * hid * 4);
    orc_rms_norm(xs, m->out_norm, c->rms_eps, B, hid, xn);
    ORC_MM(m->out.w, m->out.type, m->out.n, m->out.k, xn, B, logits, o2);
    g_phase[2] += (orc_now() - t_step) - (g_phase[0] - mm0) - (g_phase[1] - at0);
    free(xs); free(xn); free(q); free(k); free(v); free(att); free(g); free(u); free(tmp);
}

/* Borrowed view of one weight tensor (native GGUF blocks) so a test can hand the SAME bytes to the GPU path.
 * layer < 0: the output matrix. */
const uint8_t* orc_llama_get_qweight(void* mp, int layer, int which, int* type, int* n, int* k) {
    orc_model* m = (orc_model*)mp;
    const orc_qw* q = layer < 0 ? &m->out : &m->lw[(size_t)layer * 7 + which];
    *type = q->type; *n = q->n; *k = q->k;
    return q->w;
}

/* One prompt step of ONE sequence of T tokens without a cached prefix (`is_prefill`, quantized_llama.rs:424-506 with
 * the causal NaiveAttention of models/mod.rs:1288-1306 and the mask of layers/mask.rs:32-53): same op order and
 * rounding points as orc_llama_decode / oracle/llama.py (q, k, v rounded to bf16 before attention, attention output
 * rounded to bf16, f32 everywhere else), quantised products in O1f.  Writes the chunk's K/V into the flash-layout
 * caches at `slots` and returns the LAST token's logits [vocab]. */
void orc_llama_prefill(void* mp, const uint32_t* tokens, const int64_t* positions, const int64_t* slots, int T,
                       uint16_t** kcache, uint16_t** vcache, float* logits) {
    orc_model* m = (orc_model*)mp;
    const orc_cfg* c = &m->c;
    const int H = c->n_heads, Hkv = c->n_kv_heads, D = c->head_dim, hid = c->hidden, I = c->intermediate;
    const int G = H / Hkv;
    float* xs = (float*)malloc((size_t)T * hid * 4);
    float* xn = (float*)malloc((size_t)T * hid * 4);
    float* q = (float*)malloc((size_t)T * H * D * 4);
    float* k = (float*)malloc((size_t)T * Hkv * D * 4);
    float* v = (float*)malloc((size_t)T * Hkv * D * 4);
    float* att = (float*)malloc((size_t)T * H * D * 4);
    float* g = (float*)malloc((size_t)T * I * 4);
    float* u = (float*)malloc((size_t)T * I * 4);
    float* tmp = (float*)malloc((size_t)T * hid * 4);
    for (int t = 0; t < T; ++t)
        for (int i = 0; i < hid; ++i)
            xs[(size_t)t * hid + i] = m->tok_embd ? m->tok_embd[(size_t)tokens[t] * hid + i]
                                                  : 0.02f * sinf((float)(tokens[t] % 977) * 0.37f + (float)i * 0.011f);
    const float scale = 1.0f / sqrtf((float)D);
    for (int l = 0; l < c->n_layers; ++l) {
        const orc_qw* W = m->lw + (size_t)l * 7;
        orc_rms_norm(xs, m->norms[(size_t)l * 2], c->rms_eps, T, hid, xn);
        orc_qmatmul(W[0].w, W[0].type, W[0].n, W[0].k, xn, T, q, 2);
        orc_qmatmul(W[1].w, W[1].type, W[1].n, W[1].k, xn, T, k, 2);
        orc_qmatmul(W[2].w, W[2].type, W[2].n, W[2].k, xn, T, v, 2);
        orc_rope_i(q, m->cosT, m->sinT, positions, T, H, D);
        orc_rope_i(k, m->cosT, m->sinT, positions, T, Hkv, D);
        for (size_t i = 0; i < (size_t)T * H * D; ++i) q[i] = round_bf16(q[i]);
        for (size_t i = 0; i < (size_t)T * Hkv * D; ++i) { k[i] = round_bf16(k[i]); v[i] = round_bf16(v[i]); }
        for (int t = 0; t < T; ++t) {
            if (slots[t] < 0) continue;
            for (int i = 0; i < Hkv * D; ++i) {
                kcache[l][(size_t)slots[t] * Hkv * D + i] = f32_to_bf16(k[(size_t)t * Hkv * D + i]);
                vcache[l][(size_t)slots[t] * Hkv * D + i] = f32_to_bf16(v[(size_t)t * Hkv * D + i]);
            }
        }
#pragma omp parallel for collapse(2) schedule(dynamic, 8)
        for (int h = 0; h < H; ++h)
            for (int t = 0; t < T; ++t) {
                const int hk = h / G;
                float* sc = (float*)malloc((size_t)(t + 1) * 4);
                const float* qr = q + ((size_t)t * H + h) * D;
                float mx = -1e30f;
                for (int j = 0; j <= t; ++j) {                 /* causal: query t sees keys 0..t */
                    const float* kr = k + ((size_t)j * Hkv + hk) * D;
                    float s = 0;
                    for (int d = 0; d < D; ++d) s += qr[d] * kr[d];
                    sc[j] = s * scale;
                    if (sc[j] > mx) mx = sc[j];
                }
                double den = 0;
                for (int j = 0; j <= t; ++j) { sc[j] = expf(sc[j] - mx); den += sc[j]; }
                float o[512];
                for (int d = 0; d < D; ++d) o[d] = 0;
                for (int j = 0; j <= t; ++j) {
                    const float* vr = v + ((size_t)j * Hkv + hk) * D;
                    const float p = (float)(sc[j] / den);
                    for (int d = 0; d < D; ++d) o[d] += p * vr[d];
                }
                float* op = att + ((size_t)t * H + h) * D;
                for (int d = 0; d < D; ++d) op[d] = round_bf16(o[d]);
                free(sc);
            }
        orc_qmatmul(W[3].w, W[3].type, W[3].n, W[3].k, att, T, tmp, 2);
        for (size_t i = 0; i < (size_t)T * hid; ++i) xs[i] += tmp[i];
        orc_rms_norm(xs, m->norms[(size_t)l * 2 + 1], c->rms_eps, T, hid, xn);
        orc_qmatmul(W[4].w, W[4].type, W[4].n, W[4].k, xn, T, g, 2);
        orc_qmatmul(W[6].w, W[6].type, W[6].n, W[6].k, xn, T, u, 2);
        for (size_t i = 0; i < (size_t)T * I; ++i) g[i] = g[i] / (1.f + expf(-g[i])) * u[i];
        orc_qmatmul(W[5].w, W[5].type, W[5].n, W[5].k, g, T, tmp, 2);
        for (size_t i = 0; i < (size_t)T * hid; ++i) xs[i] += tmp[i];
    }
    orc_rms_norm(xs + (size_t)(T - 1) * hid, m->out_norm, c->rms_eps, 1, hid, xn);
    orc_qmatmul(m->out.w, m->out.type, m->out.n, m->out.k, xn, 1, logits, 0);
    free(xs); free(xn); free(q); free(k); free(v); free(att); free(g); free(u); free(tmp);
}

/* y[N] = W[N,K] (bf16 bit patterns, row-major) . x[K] (f32), f32 accumulation: the mat-vec a 16-bit `Linear::forward`
 * (src/openai/models/linear.rs:124-172) runs on candle's CPU backend for one decode token [EXT: candle converts bf16 to f32
 * lanes and accumulates in f32].  Used by bench.py's configs[0] CPU baseline (StableLM-3B bf16 shapes, stable_lm.rs:158-212). */
void orc_bf16_gemv(const uint16_t* w, int N, int K, const float* x, float* y) {
#pragma omp parallel for schedule(static)
    for (int n = 0; n < N; ++n) {
        const uint16_t* r = w + (size_t)n * K;
        float acc = 0.f;
        int k = 0;
#if defined(__AVX2__) && defined(__FMA__)
        __m256 a0 = _mm256_setzero_ps(), a1 = _mm256_setzero_ps();
        for (; k + 16 <= K; k += 16) {
            const __m256i lo = _mm256_slli_epi32(_mm256_cvtepu16_epi32(_mm_loadu_si128((const __m128i*)(r + k))), 16);
            const __m256i hi = _mm256_slli_epi32(_mm256_cvtepu16_epi32(_mm_loadu_si128((const __m128i*)(r + k + 8))), 16);
            a0 = _mm256_fmadd_ps(_mm256_castsi256_ps(lo), _mm256_loadu_ps(x + k), a0);
            a1 = _mm256_fmadd_ps(_mm256_castsi256_ps(hi), _mm256_loadu_ps(x + k + 8), a1);
        }
        a0 = _mm256_add_ps(a0, a1);
        __m128 s = _mm_add_ps(_mm256_castps256_ps128(a0), _mm256_extractf128_ps(a0, 1));
        s = _mm_add_ps(s, _mm_movehl_ps(s, s));
        s = _mm_add_ss(s, _mm_shuffle_ps(s, s, 1));
        acc = _mm_cvtss_f32(s);
#endif
        for (; k < K; ++k) acc += bf16_to_f32(r[k]) * x[k];
        y[n] = acc;
    }
}

/* bench.py's cpu_baseline picks the thread count that gives the fastest step on the host it runs on */
void orc_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
