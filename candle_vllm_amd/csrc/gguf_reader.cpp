// GGUF on-disk format reader: header, metadata key/values, tensor directory, mmap'ed tensor data.
// What the reference reads through candle's `gguf_file::Content` (src/backend/gguf.rs:48-102,623-712;
// layers/quantized_var_builder.rs:27-58) and the keys / tensor names GGUFLLaMa consumes
// (src/openai/models/quantized_llama.rs:225-371).  Format: magic "GGUF", version 2/3 (little endian), u64 tensor
// and kv counts, then kvs {string key, u32 type, value}, tensor infos {string name, u32 n_dims, u64 dims[] (fastest
// first), u32 ggml type, u64 offset}, then padding to `general.alignment` (default 32) and the data section.  [EXT: the
// published GGUF specification]
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <algorithm>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "../../include/mi355_vllm.h"

namespace {

enum { T_U8 = 0, T_I8, T_U16, T_I16, T_U32, T_I32, T_F32, T_BOOL, T_STRING, T_ARRAY, T_U64, T_I64, T_F64 };

struct Value { uint32_t type = 0; uint64_t u = 0; int64_t i = 0; double f = 0; std::string s; uint32_t arr_type = 0; uint64_t arr_len = 0; };
struct TInfo { std::string name; uint32_t n_dims = 0; uint64_t dims[4] = {1, 1, 1, 1}; uint32_t type = 0; uint64_t offset = 0, nbytes = 0; };

struct Gguf {
    int fd = -1; const uint8_t* base = nullptr; size_t size = 0;
    uint32_t version = 0; uint64_t data_off = 0; uint32_t alignment = 32;
    std::unordered_map<std::string, Value> kv;
    std::vector<TInfo> tensors;
    std::unordered_map<std::string, int> index;
};

struct Cur {
    const uint8_t* p; const uint8_t* end; bool ok = true;
    template <typename T> T rd() { T v{}; if (!ok || sizeof(T) > (size_t)(end - p)) { ok = false; return v; } memcpy(&v, p, sizeof(T)); p += sizeof(T); return v; }
    // lengths come from the file: compare against the bytes that are left, never form p + n (it may wrap)
    std::string str() { const uint64_t n = rd<uint64_t>(); if (!ok || n > (uint64_t)(end - p)) { ok = false; return {}; } std::string s((const char*)p, n); p += n; return s; }
};

// bytes of `n` elements of ggml type t (block formats: elements per block / bytes per block)
bool type_layout(uint32_t t, uint64_t* elems_per_block, uint64_t* bytes_per_block) {
    switch (t) {
        case 0: *elems_per_block = 1; *bytes_per_block = 4; return true;      // F32
        case 1: *elems_per_block = 1; *bytes_per_block = 2; return true;      // F16
        case 2: *elems_per_block = 32; *bytes_per_block = 18; return true;    // Q4_0
        case 3: *elems_per_block = 32; *bytes_per_block = 20; return true;    // Q4_1
        case 6: *elems_per_block = 32; *bytes_per_block = 22; return true;    // Q5_0
        case 7: *elems_per_block = 32; *bytes_per_block = 24; return true;    // Q5_1
        case 8: *elems_per_block = 32; *bytes_per_block = 34; return true;    // Q8_0
        case 10: *elems_per_block = 256; *bytes_per_block = 84; return true;  // Q2_K
        case 11: *elems_per_block = 256; *bytes_per_block = 110; return true; // Q3_K
        case 12: *elems_per_block = 256; *bytes_per_block = 144; return true; // Q4_K
        case 13: *elems_per_block = 256; *bytes_per_block = 176; return true; // Q5_K
        case 14: *elems_per_block = 256; *bytes_per_block = 210; return true; // Q6_K
        case 15: *elems_per_block = 256; *bytes_per_block = 292; return true; // Q8_K
        case 30: *elems_per_block = 1; *bytes_per_block = 2; return true;     // BF16
        default: return false;
    }
}

bool read_value(Cur& c, uint32_t type, Value* v, int depth = 0) {
    v->type = type;
    switch (type) {
        case T_U8: v->u = c.rd<uint8_t>(); break;
        case T_I8: v->i = c.rd<int8_t>(); v->u = (uint64_t)v->i; break;
        case T_U16: v->u = c.rd<uint16_t>(); break;
        case T_I16: v->i = c.rd<int16_t>(); v->u = (uint64_t)v->i; break;
        case T_U32: v->u = c.rd<uint32_t>(); break;
        case T_I32: v->i = c.rd<int32_t>(); v->u = (uint64_t)v->i; break;
        case T_F32: v->f = c.rd<float>(); break;
        case T_BOOL: v->u = c.rd<uint8_t>(); break;
        case T_STRING: v->s = c.str(); break;
        case T_U64: v->u = c.rd<uint64_t>(); break;
        case T_I64: v->i = c.rd<int64_t>(); v->u = (uint64_t)v->i; break;
        case T_F64: v->f = c.rd<double>(); break;
        case T_ARRAY: {
            if (depth > 0) return false;
            v->arr_type = c.rd<uint32_t>(); v->arr_len = c.rd<uint64_t>();
            for (uint64_t k = 0; k < v->arr_len && c.ok; ++k) { Value e; if (!read_value(c, v->arr_type, &e, 1)) return false; }   // skipped (tokenizer tables)
            break;
        }
        default: return false;
    }
    if (type != T_I8 && type != T_I16 && type != T_I32 && type != T_I64) v->i = (int64_t)v->u;
    return c.ok;
}

Gguf* G(void* p) { return static_cast<Gguf*>(p); }

}  // namespace

extern "C" {

void mi355_gguf_close(void* h);

static void* gguf_open_impl(const char* path) {
    if (!path) return nullptr;
    Gguf* g = new Gguf();
    struct Guard { Gguf* g; ~Guard() { if (g) mi355_gguf_close(g); } } guard{g};     // every early exit and every throw cleans up
    g->fd = open(path, O_RDONLY);
    struct stat st;
    if (g->fd < 0 || fstat(g->fd, &st) != 0) return nullptr;
    g->size = (size_t)st.st_size;
    if (g->size < 24) return nullptr;                                       // magic + version + two counts
    void* m = mmap(nullptr, g->size, PROT_READ, MAP_PRIVATE, g->fd, 0);
    if (m == MAP_FAILED) return nullptr;
    g->base = static_cast<const uint8_t*>(m);
    Cur c{g->base, g->base + g->size};
    const uint32_t magic = c.rd<uint32_t>();
    g->version = c.rd<uint32_t>();
    bool ok = c.ok && magic == 0x46554747u && (g->version == 2 || g->version == 3);   // "GGUF"
    uint64_t n_tensors = 0, n_kv = 0;
    if (ok) { n_tensors = c.rd<uint64_t>(); n_kv = c.rd<uint64_t>(); ok = c.ok && n_tensors < (1u << 24) && n_kv < (1u << 24); }
    for (uint64_t i = 0; ok && i < n_kv; ++i) {
        const std::string key = c.str();
        const uint32_t type = c.rd<uint32_t>();
        Value v;
        ok = c.ok && read_value(c, type, &v);
        if (ok) g->kv[key] = v;
    }
    for (uint64_t i = 0; ok && i < n_tensors; ++i) {
        TInfo t;
        t.name = c.str();
        t.n_dims = c.rd<uint32_t>();
        ok = c.ok && t.n_dims >= 1 && t.n_dims <= 4;
        for (uint32_t d = 0; ok && d < t.n_dims; ++d) t.dims[d] = c.rd<uint64_t>();
        t.type = c.rd<uint32_t>();
        t.offset = c.rd<uint64_t>();
        uint64_t epb = 0, bpb = 0;
        ok = ok && c.ok && type_layout(t.type, &epb, &bpb);
        if (ok) {
            // element and byte counts with overflow checks: the dims are untrusted, and every later slice trusts nbytes
            uint64_t n = 1;
            for (uint32_t d = 0; ok && d < t.n_dims; ++d) ok = !__builtin_mul_overflow(n, t.dims[d], &n);
            ok = ok && (t.dims[0] % epb) == 0 && !__builtin_mul_overflow(n / epb, bpb, &t.nbytes);
        }
        if (ok) {
            g->index[t.name] = (int)g->tensors.size();
            g->tensors.push_back(t);
        }
    }
    if (ok) {
        auto al = g->kv.find("general.alignment");
        if (al != g->kv.end()) {
            if (al->second.u == 0 || al->second.u > (1u << 20)) ok = false;      // a zero or absurd alignment is a broken file
            else g->alignment = (uint32_t)al->second.u;
        }
        const uint64_t pos = (uint64_t)(c.p - g->base);
        g->data_off = (pos + g->alignment - 1) / g->alignment * g->alignment;
        ok = ok && g->data_off <= g->size;
        const uint64_t room = ok ? g->size - g->data_off : 0;
        for (auto& t : g->tensors) if (t.offset > room || t.nbytes > room - t.offset) ok = false;
    }
    if (!ok) return nullptr;
    guard.g = nullptr;
    return g;
}
void* mi355_gguf_open(const char* path) {
    try { return gguf_open_impl(path); } catch (...) { return nullptr; }      // bad_alloc / length_error never cross the C ABI
}
void mi355_gguf_close(void* h) {
    Gguf* g = G(h);
    if (!g) return;
    if (g->base) munmap(const_cast<uint8_t*>(g->base), g->size);
    if (g->fd >= 0) close(g->fd);
    delete g;
}
int32_t mi355_gguf_version(void* h) { return h ? (int32_t)G(h)->version : -1; }
int32_t mi355_gguf_n_tensors(void* h) { return h ? (int32_t)G(h)->tensors.size() : -1; }
/* metadata: returns 1 when the key exists with a compatible type */
int32_t mi355_gguf_get_u64(void* h, const char* key, uint64_t* out) {
    if (!h || !key || !out) return 0;
    auto it = G(h)->kv.find(key);
    if (it == G(h)->kv.end() || it->second.type == T_STRING || it->second.type == T_ARRAY || it->second.type == T_F32 || it->second.type == T_F64) return 0;
    *out = it->second.u; return 1;
}
int32_t mi355_gguf_get_f64(void* h, const char* key, double* out) {
    if (!h || !key || !out) return 0;
    auto it = G(h)->kv.find(key);
    if (it == G(h)->kv.end() || (it->second.type != T_F32 && it->second.type != T_F64)) return 0;
    *out = it->second.f; return 1;
}
int32_t mi355_gguf_get_str(void* h, const char* key, char* out, int32_t cap) {
    if (!h || !key) return -1;
    auto it = G(h)->kv.find(key);
    if (it == G(h)->kv.end() || it->second.type != T_STRING) return -1;
    const int n = (int)it->second.s.size();
    if (out && cap > 0) { const int k = n < cap - 1 ? n : cap - 1; memcpy(out, it->second.s.data(), k); out[k] = 0; }
    return n;
}
int32_t mi355_gguf_find(void* h, const char* name) { if (!h || !name) return -1; auto it = G(h)->index.find(name); return it == G(h)->index.end() ? -1 : it->second; }
/* dims are returned slowest-first ([rows, cols] for a matrix, as candle reports shapes; GGUF stores fastest-first) */
int32_t mi355_gguf_tensor_info(void* h, int32_t i, char* name, int32_t name_cap, int64_t* dims4, int32_t* n_dims, int32_t* ggml_type,
                               uint64_t* nbytes) {
    Gguf* g = G(h);
    if (!g || i < 0 || i >= (int)g->tensors.size()) return -1;
    const TInfo& t = g->tensors[i];
    if (name && name_cap > 0) { const int k = (int)t.name.size() < name_cap - 1 ? (int)t.name.size() : name_cap - 1; memcpy(name, t.name.data(), k); name[k] = 0; }
    if (dims4) for (uint32_t d = 0; d < 4; ++d) dims4[d] = d < t.n_dims ? (int64_t)t.dims[t.n_dims - 1 - d] : 1;
    if (n_dims) *n_dims = (int32_t)t.n_dims;
    if (ggml_type) *ggml_type = (int32_t)t.type;
    if (nbytes) *nbytes = t.nbytes;
    return 0;
}
const void* mi355_gguf_tensor_data(void* h, int32_t i) {
    Gguf* g = G(h);
    if (!g || i < 0 || i >= (int)g->tensors.size()) return nullptr;
    return g->base + g->data_off + g->tensors[i].offset;
}
/* Tensor-parallel shard of tensor i as a raw byte range (no dequantisation): what `QVarBuilder::get_sharded` asks of
 * candle's `Content::tensor_shard` (layers/quantized_var_builder.rs:135-183).  dim 0 = a contiguous range of rows
 * (column-parallel weights, `shard(0, rank, W)`), dim 1 = the same block range of every row (row-parallel weights).
 * Returns the shard's byte count (`out` may be NULL to size the buffer); -1 for a bad argument or a dimension that does
 * not divide; -2 when a dim-1 shard would cut a quantisation block -- the reference then dequantises, narrows and
 * re-quantises to Q8_0 (quantized_var_builder.rs:234-269), which changes the weights and is not built here;
 * -3 when `out_cap` is too small. */
int64_t mi355_gguf_tensor_shard(void* h, int32_t i, int32_t dim, int32_t rank, int32_t world, void* out, int64_t out_cap) {
    Gguf* g = G(h);
    if (!g || i < 0 || i >= (int)g->tensors.size() || world < 1 || rank < 0 || rank >= world) return -1;
    const TInfo& t = g->tensors[i];
    if (t.n_dims > 2 || dim < 0 || dim >= (int)t.n_dims) return -1;
    uint64_t epb = 0, bpb = 0;
    if (!type_layout(t.type, &epb, &bpb)) return -1;
    const uint64_t cols = t.dims[0], rows = t.n_dims == 2 ? t.dims[1] : 1;      // GGUF stores the fastest dimension first
    const uint64_t row_bytes = cols / epb * bpb;
    const uint8_t* src = g->base + g->data_off + t.offset;
    uint8_t* dst = static_cast<uint8_t*>(out);
    const bool split_rows = t.n_dims == 2 && dim == 0;
    if (split_rows) {
        if (rows % (uint64_t)world) return -1;
        const uint64_t n = rows / world * row_bytes;
        if (dst) {
            if (out_cap < (int64_t)n) return -3;
            memcpy(dst, src + (uint64_t)rank * n, n);
        }
        return (int64_t)n;
    }
    // split along the fastest dimension (dim 1 of a matrix, dim 0 of a vector)
    if (cols % (uint64_t)world) return -1;
    const uint64_t c = cols / world;
    if (c % epb) return -2;
    const uint64_t seg = c / epb * bpb, n = seg * rows;
    if (dst) {
        if (out_cap < (int64_t)n) return -3;
        for (uint64_t r = 0; r < rows; ++r) memcpy(dst + r * seg, src + r * row_bytes + (uint64_t)rank * seg, seg);
    }
    return (int64_t)n;
}

/* rows [row0, row0 + n_rows) of a 2-D tensor in the file's own block format; rows beyond the tensor come out as ZERO blocks (all bytes
 * zero: d = dmin = 0 in Q4_K / Q6_K / Q8_0, i.e. rows of 0.0) -- the vocab-parallel lm_head of a padded vocabulary (VocabParallelLinear,
 * distributed.rs:1596-1616).  Returns the byte count (out == NULL: size query); -1 bad arguments / type, -3 out_cap too small. */
int64_t mi355_gguf_tensor_rows_padded(void* h, int32_t i, int64_t row0, int64_t n_rows, void* out, int64_t out_cap) {
    Gguf* g = G(h);
    if (!g || i < 0 || i >= (int)g->tensors.size() || row0 < 0 || n_rows <= 0) return -1;
    const TInfo& t = g->tensors[i];
    if (t.n_dims != 2) return -1;
    uint64_t epb = 0, bpb = 0;
    if (!type_layout(t.type, &epb, &bpb)) return -1;
    const uint64_t cols = t.dims[0], rows = t.dims[1];
    if (cols % epb) return -1;
    const uint64_t row_bytes = cols / epb * bpb;
    if ((uint64_t)n_rows > (uint64_t)INT64_MAX / (row_bytes ? row_bytes : 1)) return -1;
    const uint64_t n = (uint64_t)n_rows * row_bytes;
    if (!out) return (int64_t)n;
    if (out_cap < (int64_t)n) return -3;
    const uint64_t have = (uint64_t)row0 < rows ? std::min<uint64_t>((uint64_t)n_rows, rows - (uint64_t)row0) : 0;
    uint8_t* dst = static_cast<uint8_t*>(out);
    if (have) memcpy(dst, g->base + g->data_off + t.offset + (uint64_t)row0 * row_bytes, have * row_bytes);
    memset(dst + have * row_bytes, 0, (size_t)(n - have * row_bytes));
    return (int64_t)n;
}

// IEEE binary16 <-> binary32 in plain integer arithmetic (round to nearest even, subnormals kept): the host side must not
// depend on a compiler's _Float16 (g++ has none in C++ mode; the sanitizer build uses g++)
static inline float half_bits_to_float(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, exp = (h >> 10) & 0x1Fu, man = h & 0x3FFu;
    uint32_t u;
    if (exp == 0) {
        if (man == 0) { u = sign; }
        else {                                                    // subnormal: normalise
            int e = -1;
            uint32_t m = man;
            do { ++e; m <<= 1; } while (!(m & 0x400u));
            u = sign | ((uint32_t)(127 - 15 - e) << 23) | ((m & 0x3FFu) << 13);
        }
    } else if (exp == 31) { u = sign | 0x7F800000u | (man << 13); }
    else { u = sign | ((exp + 112u) << 23) | (man << 13); }
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static inline uint16_t float_to_half_bits(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    const uint32_t sign = (u >> 16) & 0x8000u;
    u &= 0x7FFFFFFFu;
    if (u >= 0x7F800000u) return (uint16_t)(sign | 0x7C00u | (u > 0x7F800000u ? 0x200u : 0u));   // inf / nan
    if (u >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u);                                        // rounds to inf (>= 65520)
    if (u < 0x33000001u) return (uint16_t)sign;                                                     // <= 2^-25: rounds to zero
    const int e = (int)(u >> 23) - 127;
    uint32_t man = (u & 0x7FFFFFu) | 0x800000u;
    int shift = e < -14 ? 13 + (-14 - e) : 13;                                                      // subnormal results lose more bits
    uint32_t hm = man >> shift;
    const uint32_t rem = man & ((1u << shift) - 1u), halfway = 1u << (shift - 1);
    if (rem > halfway || (rem == halfway && (hm & 1u))) ++hm;
    const uint32_t he = e < -14 ? 0u : (uint32_t)(e + 15);
    // hm carries the implicit bit for normal results: (he << 10) + (hm - 0x400) == ((he - 1) << 10) + hm; a mantissa carry rolls into the exponent
    const uint32_t out = e < -14 ? hm : (((he - 1u) << 10) + hm);
    return (uint16_t)(sign | out);
}


/* The reference's fallback for a dim-1 shard that cuts a quantisation block (`get_sharded_no_shape`,
 * layers/quantized_var_builder.rs:234-269): dequantise the WHOLE tensor to f16 (`dequantize_f16`: f32 arithmetic of the block
 * format, one rounding to f16), narrow columns [rank*c, (rank+1)*c), re-quantise -- to Q8_0 when c is not a multiple of the
 * source block (256), which is the only case that reaches this function here.  Q8_0 as ggml's `quantize_row_q8_0` [EXT]:
 * d = amax / 127 in f32, q = roundf(x / d) with the UNROUNDED d, stored d rounded to f16.
 * out: rows x (c / 32) blocks of 34 bytes.  Returns the byte count (out may be NULL); -1 bad argument / unsupported source
 * type (Q4_K and Q6_K are restated here) / c % 32 != 0; -3 out_cap too small. */
int64_t mi355_gguf_tensor_shard_q8_0(void* h, int32_t i, int32_t rank, int32_t world, void* out, int64_t out_cap) {
#pragma clang fp contract(off)      // candle's dequantisation rounds the product and the difference separately (no fused multiply-add)
    Gguf* g = G(h);
    if (!g || i < 0 || i >= (int)g->tensors.size() || world < 1 || rank < 0 || rank >= world) return -1;
    const TInfo& t = g->tensors[i];
    if (t.n_dims != 2 || (t.type != 12 && t.type != 14)) return -1;
    const uint64_t cols = t.dims[0], rows = t.dims[1];
    if (cols % (uint64_t)world) return -1;
    const uint64_t c = cols / world;
    if (c % 32) return -1;
    const uint64_t n = rows * (c / 32) * 34;
    if (!out) return (int64_t)n;
    if (out_cap < (int64_t)n) return -3;
    const uint64_t bb = t.type == 12 ? 144 : 210, nb = cols / 256;
    const uint8_t* src = g->base + g->data_off + t.offset;
    uint8_t* dst = static_cast<uint8_t*>(out);
    auto f16 = [](const uint8_t* p) { uint16_t u; memcpy(&u, p, 2); return half_bits_to_float(u); };
    try {
        std::vector<float> row(cols);
#pragma omp parallel for schedule(static) firstprivate(row)
        for (int64_t r = 0; r < (int64_t)rows; ++r) {
            const uint8_t* rp = src + (uint64_t)r * nb * bb;
            for (uint64_t b = 0; b < nb; ++b) {                       // ggml dequantize_row_q4_K / q6_K, f32 arithmetic
                const uint8_t* blk = rp + b * bb;
                float* y = row.data() + b * 256;
                if (t.type == 12) {
                    const float d = f16(blk), dmin = f16(blk + 2);
                    const uint8_t* sc = blk + 4;
                    const uint8_t* q = blk + 16;
                    int is = 0;
                    for (int j = 0; j < 256; j += 64, q += 32, is += 2) {
                        uint8_t s1, m1, s2, m2;
                        auto sm = [&](int jj, uint8_t& S, uint8_t& M) {
                            if (jj < 4) { S = sc[jj] & 63; M = sc[jj + 4] & 63; }
                            else { S = (sc[jj + 4] & 0xF) | ((sc[jj - 4] >> 6) << 4); M = (sc[jj + 4] >> 4) | ((sc[jj] >> 6) << 4); }
                        };
                        sm(is, s1, m1); sm(is + 1, s2, m2);
                        const float d1 = d * s1, mm1 = dmin * m1, d2 = d * s2, mm2 = dmin * m2;
                        for (int l = 0; l < 32; ++l) y[j + l] = d1 * (q[l] & 0xF) - mm1;
                        for (int l = 0; l < 32; ++l) y[j + 32 + l] = d2 * (q[l] >> 4) - mm2;
                    }
                } else {
                    const uint8_t *ql = blk, *qh = blk + 128;
                    const int8_t* sc = reinterpret_cast<const int8_t*>(blk + 192);
                    const float d = f16(blk + 208);
                    for (int nh = 0; nh < 2; ++nh, y += 128, ql += 64, qh += 32, sc += 8)
                        for (int l = 0; l < 32; ++l) {
                            const int isx = l / 16;
                            const int q1 = (int)((ql[l] & 0xF) | (((qh[l] >> 0) & 3) << 4)) - 32;
                            const int q2 = (int)((ql[l + 32] & 0xF) | (((qh[l] >> 2) & 3) << 4)) - 32;
                            const int q3 = (int)((ql[l] >> 4) | (((qh[l] >> 4) & 3) << 4)) - 32;
                            const int q4 = (int)((ql[l + 32] >> 4) | (((qh[l] >> 6) & 3) << 4)) - 32;
                            y[l] = d * sc[isx] * q1; y[l + 32] = d * sc[isx + 2] * q2;
                            y[l + 64] = d * sc[isx + 4] * q3; y[l + 96] = d * sc[isx + 6] * q4;
                        }
                }
            }
            uint8_t* orow = dst + (uint64_t)r * (c / 32) * 34;
            const float* x = row.data() + (uint64_t)rank * c;
            for (uint64_t b = 0; b < c / 32; ++b) {
                float v[32], amax = 0.f;
                for (int j = 0; j < 32; ++j) { v[j] = half_bits_to_float(float_to_half_bits(x[b * 32 + j])); amax = fmaxf(amax, fabsf(v[j])); }   // dequantize_f16: one rounding
                const float d = amax / 127.f, id = d != 0.f ? 1.f / d : 0.f;
                const uint16_t dh = float_to_half_bits(d);
                memcpy(orow + b * 34, &dh, 2);
                for (int j = 0; j < 32; ++j) orow[b * 34 + 2 + j] = (uint8_t)(int8_t)roundf(v[j] * id);
            }
        }
    } catch (...) { return -1; }
    return (int64_t)n;
}

}  // extern "C"
