// RoPE cos/sin table builders (HOST code): the reference's `DefaultRotaryEmbedding::new` and every arm of
// `ScalingRotaryEmbedding::new` (src/openai/models/layers/rotary_emb.rs:14-48,107-341,358-457) restated in the
// reference's arithmetic -- inverse frequencies rounded to f32, angle = f32(position) * f32(inv_freq), cos / sin of
// that f32 angle -- so the tables agree with a candle build to the last bit of libm.  The tables are f32
// [n_positions, rotary_dim / 2]; the kernels (fused QKV epilogue, mi355_rope_inplace) only ever read them.
#include <cmath>
#include <cstdint>
#include <vector>

#include "../../include/mi355_vllm.h"

namespace {

// calculate_default_inv_freq (rotary_emb.rs:14-19): base^(i/dim) in f64, reciprocal taken in f32
std::vector<float> default_inv_freq(double base, int dim) {
    std::vector<float> f;
    for (int i = 0; i < dim; i += 2) f.push_back(1.0f / (float)pow(base, (double)i / (double)dim));
    return f;
}

void fill(float* cos_out, float* sin_out, const std::vector<float>& inv_freq, int n_positions, float pos_div, float mscale) {
    const int half = (int)inv_freq.size();
    for (int p = 0; p < n_positions; ++p) {
        const float t = pos_div == 1.0f ? (float)p : (float)((double)(float)p / (double)pos_div);   // linear: arange / factor (f32 tensor / f64 scalar)
        for (int i = 0; i < half; ++i) {
            const float th = t * inv_freq[i];
            float c = cosf(th), s = sinf(th);
            if (mscale != 1.0f) { c = (float)((double)c * (double)mscale); s = (float)((double)s * (double)mscale); }
            cos_out[(size_t)p * half + i] = c;
            sin_out[(size_t)p * half + i] = s;
        }
    }
}

double original_max(const mi355_rope_scaling* sc, int max_position_embeddings, int max_seq_len) {
    // rotary_emb.rs:123-135
    if (sc->original_max_position_embeddings > 0) return sc->original_max_position_embeddings;
    if (sc->factor > 0 && max_position_embeddings > 0) return (double)max_position_embeddings / sc->factor;
    return max_position_embeddings > 0 ? max_position_embeddings : max_seq_len;
}

float yarn_correction_dim(float num_rot, int dim, float base, int max_pos) {     // rotary_emb.rs:359-367
    return ((float)dim * logf((float)max_pos / (num_rot * 2.f * (float)M_PI))) / (2.f * logf(base));
}

}  // namespace

extern "C" int32_t mi355_rope_table_len(const mi355_rope_scaling* sc, int32_t max_seq_len, int32_t max_position_embeddings) {
    if (!sc || sc->type == MI355_ROPE_DEFAULT || sc->type == MI355_ROPE_LLAMA3) return max_seq_len;
    const double orig = original_max(sc, max_position_embeddings, max_seq_len);
    if (sc->type == MI355_ROPE_LINEAR) return (int32_t)(uint32_t)(orig * sc->factor);
    if (sc->type == MI355_ROPE_DYNAMIC) {
        if (sc->alpha > 0) return max_position_embeddings;
        return (int32_t)(uint32_t)(orig * sc->factor);
    }
    if (sc->type == MI355_ROPE_YARN) {
        const int mp = max_position_embeddings > 0 ? max_position_embeddings : max_seq_len;
        return (int32_t)(uint32_t)((float)mp * (float)sc->factor);
    }
    return -1;
}

// Config::effective_max_seq_len (src/openai/models/mod.rs:663-702): the context length a yarn-scaled model serves,
// max(base, round(original_max_position_embeddings * factor)) for factor > 1; every other scaling keeps the base.
// `apply_runtime_rope_overrides(Some(f))` (:704-714) is this with {yarn, factor f, original = base}.
extern "C" int64_t mi355_effective_max_seq_len(const mi355_rope_scaling* sc, int64_t base_max_position_embeddings) {
    if (!sc || sc->type != MI355_ROPE_YARN || !(sc->factor > 1.0) || !(sc->original_max_position_embeddings > 0))
        return base_max_position_embeddings;
    const int64_t scaled = (int64_t)std::llround(sc->original_max_position_embeddings * sc->factor);
    return scaled > base_max_position_embeddings ? scaled : base_max_position_embeddings;
}

extern "C" int mi355_rope_tables(float* cos_out, float* sin_out, int32_t rotary_dim, int32_t n_positions, double rope_theta,
                                 const mi355_rope_scaling* sc, int32_t max_seq_len, int32_t max_position_embeddings) {
    if (!cos_out || !sin_out || rotary_dim <= 0 || (rotary_dim & 1) || n_positions <= 0 || rope_theta <= 0) return 1;
    const int type = sc ? sc->type : MI355_ROPE_DEFAULT;
    if (type == MI355_ROPE_DEFAULT) {
        fill(cos_out, sin_out, default_inv_freq(rope_theta, rotary_dim), n_positions, 1.0f, 1.0f);
        return 0;
    }
    const double orig = original_max(sc, max_position_embeddings, max_seq_len);
    if (type == MI355_ROPE_LINEAR) {                                  // rotary_emb.rs:138-167
        if (sc->factor <= 0) return 1;
        fill(cos_out, sin_out, default_inv_freq(rope_theta, rotary_dim), n_positions, (float)sc->factor, 1.0f);
        return 0;
    }
    if (type == MI355_ROPE_LLAMA3) {                                  // rotary_emb.rs:168-224
        if (sc->factor <= 0 || sc->low_freq_factor <= 0 || sc->high_freq_factor <= 0) return 1;
        const float low_wl = (float)(orig / sc->low_freq_factor), high_wl = (float)(orig / sc->high_freq_factor);
        std::vector<float> f = default_inv_freq(rope_theta, rotary_dim);
        for (float& freq : f) {
            const float wavelen = 2.f * (float)M_PI / freq;
            if (wavelen < high_wl) continue;
            if (wavelen > low_wl) { freq = freq / (float)sc->factor; continue; }
            const float smooth = ((float)orig / wavelen - (float)sc->low_freq_factor) / (float)(sc->high_freq_factor - sc->low_freq_factor);
            freq = (1.f - smooth) * freq / (float)sc->factor + smooth * freq;
        }
        fill(cos_out, sin_out, f, n_positions, 1.0f, 1.0f);
        return 0;
    }
    if (type == MI355_ROPE_DYNAMIC) {                                 // rotary_emb.rs:227-277
        const double s = sc->alpha > 0 ? sc->alpha : sc->factor;
        if (s <= 0 || rotary_dim <= 2) return 1;
        double theta;
        if (sc->alpha > 0) {
            theta = pow(rope_theta * s, (double)rotary_dim / (double)(rotary_dim - 2));
        } else {
            const double max_len = (double)(uint32_t)(orig * s);
            theta = pow(rope_theta * ((s * max_len / orig) - (s - 1.0)), (double)rotary_dim / (double)(rotary_dim - 2));
        }
        fill(cos_out, sin_out, default_inv_freq(theta, rotary_dim), n_positions, 1.0f, 1.0f);
        return 0;
    }
    if (type == MI355_ROPE_YARN) {                                    // rotary_emb.rs:278-320,400-457
        if (sc->factor <= 0) return 1;
        const float factor = (float)sc->factor, base = (float)rope_theta;
        const float beta_fast = sc->beta_fast > 0 ? (float)sc->beta_fast : 32.f, beta_slow = sc->beta_slow > 0 ? (float)sc->beta_slow : 1.f;
        const float attn_factor = sc->attn_factor > 0 ? (float)sc->attn_factor : 1.f;
        const float extrapolation = sc->extrapolation_factor > 0 ? (float)sc->extrapolation_factor : 1.f;
        const int dim = rotary_dim, half = dim / 2;
        float low = floorf(yarn_correction_dim(beta_fast, dim, base, (int)orig));
        float high = ceilf(yarn_correction_dim(beta_slow, dim, base, (int)orig));
        low = fmaxf(low, 0.f);
        high = fminf(high, (float)dim - 1.f);
        if (low == high) high += 0.001f;
        std::vector<float> f(half);
        for (int k = 0; k < half; ++k) {
            const float pw = powf(base, (float)(2 * k) / (float)dim);
            const float extra = 1.f / pw, inter = 1.f / (factor * pw);
            // ramp computed by candle as f32 tensor ops with f64 scalars: (arange - min) / (max - min), clamp, 1 - x, * extrapolation
            float ramp = (float)(((double)(float)k - (double)low)) ;
            ramp = (float)((double)ramp / ((double)high - (double)low));
            ramp = fminf(fmaxf(ramp, 0.f), 1.f);
            const float mask = (float)((double)(float)(1.0 - (double)ramp) * (double)extrapolation);
            f[k] = inter * (float)(1.0 - (double)mask) + extra * mask;
        }
        const float mscale = (factor <= 1.f ? 1.f : 0.1f * 1.0f * logf(factor) + 1.f) * attn_factor;   // yarn_get_mscale(factor, 1.0) * attn_factor
        fill(cos_out, sin_out, f, n_positions, 1.0f, mscale);
        return 0;
    }
    return 1;
}

// ABI guard: struct sizes as this build of the header sees them (hand-written mirrors compare against it)
extern "C" int64_t mi355_abi_struct_size(int32_t which) {
    switch (which) {
        case 0: return (int64_t)sizeof(mi355_qmm_desc);
        case 1: return (int64_t)sizeof(mi355_llama_config);
        case 2: return (int64_t)sizeof(mi355_dense_config);
        case 3: return (int64_t)sizeof(mi355_rope_scaling);
        default: return -1;
    }
}
