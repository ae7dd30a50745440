"""Third-party anchors for the oracle (CPU only).  The reference ships no fixtures for its floating-point path
(SURVEY.md 0.5, 8c), so the restatements in oracle/ are additionally pinned to independent PUBLISHED implementations that
are importable in this image: PyTorch's CPU ops (fp8 / bf16 conversion, softmax attention, RMSNorm, SiLU) and Hugging
Face transformers' Llama / Qwen2 / StableLM modules, which candle's model code mirrors (the reference loads the same
checkpoints).  These tests pin SEMANTICS (op order, RoPE pairing, GQA mapping, causal mask, norm / bias placement, paged
decode == full attention); the rounding points are candle's, so tolerances are bf16-noise sized, and a structural slip
(wrong pairing, missing norm, off-by-one mask) shows up as an O(1) error."""
import numpy as np
import pytest

from oracle import dense_llama as DL
from oracle import kquants as kq
from oracle import llama
from oracle import ops as O

torch = pytest.importorskip("torch")
F = torch.nn.functional


# ----------------------------------------------------------------------------------------------- conversions
def test_bf16_rounding_equals_torch_on_every_bf16_neighbourhood():
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.standard_normal(200000).astype(np.float32) * 10.0 ** rng.integers(-20, 20, 200000),
                        np.array([0.0, -0.0, 1.0, 1.00390625, 1.0078125, 3.3895314e38, 1e-40, -1e-45], np.float32)])
    # exact ties: bf16 value + half an ulp
    b = rng.integers(0, 0x7F7F, 50000).astype(np.uint32) << 16
    ties = (b | 0x8000).view(np.float32)
    x = np.concatenate([x, ties, -ties])
    want = torch.from_numpy(x).to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)
    assert np.array_equal(O.f32_to_bf16_bits(x), want)


def test_e4m3fn_conversion_equals_torch_for_every_bf16_input_in_range():
    """all 2^16 bf16 bit patterns as inputs: finite values up to the format maximum convert exactly like
    torch.float8_e4m3fn (OCP e4m3fn, round to nearest even); beyond +-448 the oracle SATURATES (the cache kernels'
    behaviour) where torch produces NaN -- asserted separately; decode is checked on all 256 bytes"""
    bits = np.arange(65536, dtype=np.uint32) << 16
    x = bits.astype(np.uint32).view(np.float32)
    fin = np.isfinite(x)
    got = O.f32_to_e4m3fn(x)
    tq = torch.from_numpy(x).to(torch.float8_e4m3fn).view(torch.uint8).numpy()
    inr = fin & (np.abs(x) <= 448.0)
    assert np.array_equal(got[inr], tq[inr])
    # rounding boundary: values in (448, 464] still round to 448 in torch (464 is the tie, to even); above, torch overflows to NaN
    mid = fin & (np.abs(x) > 448.0) & (np.abs(x) <= 464.0)
    assert np.array_equal(got[mid], tq[mid]) and set(got[mid] & 0x7F) == {0x7E}
    over = fin & (np.abs(x) > 464.0)
    assert set(got[over] & 0x7F) == {0x7E} and set(tq[over] & 0x7F) == {0x7F}
    allb = np.arange(256, dtype=np.uint8)
    want = torch.from_numpy(allb).view(torch.float8_e4m3fn).to(torch.float32).numpy()
    mine = O.e4m3fn_to_f32(allb)
    assert np.array_equal(np.isnan(mine), np.isnan(want)) and np.array_equal(mine[~np.isnan(mine)], want[~np.isnan(want)])


# ----------------------------------------------------------------------------------------------- elementwise ops
def test_rms_norm_and_silu_mul_equal_torch():
    rng = np.random.default_rng(2)
    x = rng.standard_normal((5, 384)).astype(np.float32) * 3
    w = (1 + 0.1 * rng.standard_normal(384)).astype(np.float32)
    want = F.rms_norm(torch.from_numpy(x).double(), (384,), torch.from_numpy(w).double(), 1e-5).float().numpy()
    assert np.abs(O.rms_norm(x, w, 1e-5) - want).max() <= 2e-7 * np.abs(want).max()
    g, u = rng.standard_normal((4, 256)).astype(np.float32) * 4, rng.standard_normal((4, 256)).astype(np.float32)
    want = (F.silu(torch.from_numpy(g).double()) * torch.from_numpy(u).double()).float().numpy()
    assert np.abs(O.silu_mul(g, u) - want).max() <= 2e-7 * np.abs(want).max()


# ----------------------------------------------------------------------------------------------- attention
def _sdpa(q, k, v, scale, causal_offset=None):
    """q [T,H,D], k/v [S,Hkv,D] -> [T,H,D] through torch's scaled_dot_product_attention in f64"""
    H, Hkv = q.shape[1], k.shape[1]
    tq = torch.from_numpy(q).double().permute(1, 0, 2)[None]
    tk = torch.from_numpy(k).double().permute(1, 0, 2).repeat_interleave(H // Hkv, 0)[None]
    tv = torch.from_numpy(v).double().permute(1, 0, 2).repeat_interleave(H // Hkv, 0)[None]
    mask = None
    if causal_offset is not None:
        T, S = q.shape[0], k.shape[0]
        mask = torch.ones(T, S, dtype=torch.bool).tril(causal_offset)
    out = F.scaled_dot_product_attention(tq, tk, tv, attn_mask=mask, scale=scale)
    return out[0].permute(1, 0, 2).float().numpy()


@pytest.mark.parametrize("flash", [False, True])
def test_paged_decode_attention_equals_torch_sdpa(flash):
    rng = np.random.default_rng(3)
    H, Hkv, D, bs = 8, 2, 64, 16
    ctx = [37, 16, 1]
    tables = [[5, 2, 7], [1], [4]]
    ks, vs = O.kv_cache_shapes(9, bs, Hkv, D, 2, flash)
    kc, vc = np.zeros(ks, np.uint16), np.zeros(vs, np.uint16)
    q = O.round_bf16(rng.standard_normal((3, H, D)).astype(np.float32))
    dense = []
    for n, tab in zip(ctx, tables):
        k = O.round_bf16(rng.standard_normal((n, Hkv, D)).astype(np.float32))
        v = O.round_bf16(rng.standard_normal((n, Hkv, D)).astype(np.float32))
        slots = [tab[p // bs] * bs + p % bs for p in range(n)]
        O.reshape_and_cache(O.f32_to_bf16_bits(k), O.f32_to_bf16_bits(v), kc, vc, slots, flash)
        dense.append((k, v))
    bt = np.zeros((3, 3), np.uint32)
    for i, t in enumerate(tables):
        bt[i, :len(t)] = t
    got = O.paged_attention_decode(q, kc, vc, bt, ctx, 0.125, flash)
    for i, (k, v) in enumerate(dense):
        want = _sdpa(q[i:i + 1], k, v, 0.125)[0]
        assert np.abs(got[i] - want).max() <= 2.0 ** -8 * np.abs(want).max() + 1e-6      # output rounded to bf16


@pytest.mark.parametrize("cached", [0, 19])
def test_prefill_attention_equals_torch_sdpa_causal_with_cached_prefix(cached):
    rng = np.random.default_rng(4)
    T, H, Hkv, D = 23, 6, 3, 32
    q = rng.standard_normal((T, H, D)).astype(np.float32)
    k = rng.standard_normal((cached + T, Hkv, D)).astype(np.float32)
    v = rng.standard_normal((cached + T, Hkv, D)).astype(np.float32)
    got = O.prefill_attention(q, k, v, 0.2, cached=cached, rnd=lambda a: a)
    want = _sdpa(q, k, v, 0.2, causal_offset=cached)
    assert np.abs(got - want).max() <= 1e-6 * max(1.0, np.abs(want).max())


@pytest.mark.parametrize("window,cached", [(8, 0), (5, 19), (64, 3)])
def test_sliding_window_attention_equals_the_published_mask(window, cached):
    """`sliding_window` (attention.rs:566-575,888-897; layers/mask.rs:22-27 -> attention-rs, un-vendored): the oracle's window against
    torch SDPA under the mask transformers builds for Mistral-style models (`sliding_window_overlay`: kv_idx > q_idx - sliding_window,
    on top of causal) -- prompt steps with a cached prefix, and the decode step as the last query of the same sequence"""
    from transformers.masking_utils import sliding_window_overlay, causal_mask_function
    rng = np.random.default_rng(5 + window)
    T, H, Hkv, D = 23, 4, 2, 32
    S = cached + T
    q = rng.standard_normal((T, H, D)).astype(np.float32)
    k = rng.standard_normal((S, Hkv, D)).astype(np.float32)
    v = rng.standard_normal((S, Hkv, D)).astype(np.float32)
    win = sliding_window_overlay(window)
    mask = torch.tensor([[bool(win(0, 0, cached + t, j)) and bool(causal_mask_function(0, 0, cached + t, j)) for j in range(S)] for t in range(T)])
    tq = torch.from_numpy(q).double().permute(1, 0, 2)[None]
    tk = torch.from_numpy(k).double().permute(1, 0, 2).repeat_interleave(H // Hkv, 0)[None]
    tv = torch.from_numpy(v).double().permute(1, 0, 2).repeat_interleave(H // Hkv, 0)[None]
    want = F.scaled_dot_product_attention(tq, tk, tv, attn_mask=mask, scale=0.2)[0].permute(1, 0, 2).float().numpy()
    got = O.prefill_attention(q, k, v, 0.2, cached=cached, rnd=lambda a: a, sliding_window=window)
    assert np.abs(got - want).max() <= 1e-6 * max(1.0, np.abs(want).max())
    # decode: the last query over the paged cache holding all S keys == the last row of the prompt-step result
    bs = 16
    nb = -(-S // bs)
    ks, vs = O.kv_cache_shapes(nb + 1, bs, Hkv, D, 2, False)
    kc, vc = np.zeros(ks, np.uint16), np.zeros(vs, np.uint16)
    kb, vb, qb = O.round_bf16(k), O.round_bf16(v), O.round_bf16(q[-1:])
    table = list(range(nb, 0, -1))
    O.reshape_and_cache(O.f32_to_bf16_bits(kb), O.f32_to_bf16_bits(vb), kc, vc, [table[p // bs] * bs + p % bs for p in range(S)], False)
    dec = O.paged_attention_decode(qb, kc, vc, np.asarray([table], np.uint32), [S], 0.2, False, sliding_window=window)[0]
    ref = O.prefill_attention(qb, kb, vb, 0.2, cached=S - 1, sliding_window=window)[0]
    assert np.array_equal(dec, ref)


# ----------------------------------------------------------------------------------------------- whole models
def _hf_state(W, cfg, bias, norm_bias):
    sd = {"model.embed_tokens.weight": W["tok_embd"], "model.norm.weight": W["output_norm"], "lm_head.weight": W["output"]}
    if norm_bias:
        sd["model.norm.bias"] = W["output_norm_b"]
    names = {"wq": "self_attn.q_proj", "wk": "self_attn.k_proj", "wv": "self_attn.v_proj", "wo": "self_attn.o_proj",
             "w1": "mlp.gate_proj", "w3": "mlp.up_proj", "w2": "mlp.down_proj"}
    for l, lw in enumerate(W["layers"]):
        p = f"model.layers.{l}."
        for k, n in names.items():
            sd[p + n + ".weight"] = lw[k]
        if bias:
            for k, n in (("bq", "q_proj"), ("bk", "k_proj"), ("bv", "v_proj")):
                sd[p + "self_attn." + n + ".bias"] = lw[k]
        sd[p + "input_layernorm.weight"] = lw["attn_norm"]
        sd[p + "post_attention_layernorm.weight"] = lw["ffn_norm"]
        if norm_bias:
            sd[p + "input_layernorm.bias"] = lw["attn_norm_b"]
            sd[p + "post_attention_layernorm.bias"] = lw["ffn_norm_b"]
    return {k: torch.from_numpy(np.ascontiguousarray(v, np.float32)) for k, v in sd.items()}


def _hf_logits(model, sd, tokens):
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not [m for m in missing if "rotary" not in m and "inv_freq" not in m], missing
    assert not unexpected, unexpected
    model.eval()
    with torch.no_grad():
        return model(torch.tensor([tokens])).logits[0].float().numpy()


def _oracle_prefill_then_decode(orc, cfg, tokens, n_decode):
    """last-token logits of the prompt step, then n_decode decode steps fed with the given continuation"""
    bs = cfg.block_size
    prompt = len(tokens) - n_decode
    table = list(range(1, 2 + len(tokens) // bs))
    cache = orc.new_cache(len(table) + 2)
    seq = {"tokens": list(tokens[:prompt]), "block_table": table}
    rows = [orc.forward(O.prepare_prompt([seq], bs), cache, is_prefill=True)[0]]
    for i in range(n_decode):
        seq["tokens"].append(tokens[prompt + i])
        rows.append(orc.forward(O.prepare_decode([seq], bs), cache)[0])
    return np.stack(rows)


def _check_against_hf(got, want_all, first_row, tol):
    want = want_all[first_row:first_row + got.shape[0]]
    scale = np.abs(want).max()
    err = np.abs(got - want).max() / scale
    assert err < tol, err
    # the two arg-maxes agree wherever the HF margin is larger than the noise
    srt = np.sort(want, -1)
    clear = (srt[:, -1] - srt[:, -2]) > 4 * tol * scale
    assert (got.argmax(-1) == want.argmax(-1))[clear].all()
    return err


@pytest.mark.parametrize("kind", ["llama", "qwen2", "stablelm"])
def test_dense_oracle_matches_huggingface_model(kind):
    """OracleDenseLlama (candle's bf16 rounding chain) vs the HF module in f32 with the same bf16-exact weights: prompt step
    + 3 paged decode steps against HF's full-sequence forward.  Differences are bf16 rounding noise only."""
    tr = pytest.importorskip("transformers")
    if kind == "stablelm":
        cfg = DL.DenseConfig(hidden=320, n_layers=2, n_heads=4, n_kv_heads=4, head_dim=80, intermediate=512, vocab=384,
                             rope_theta=10000.0, max_seq=128, block_size=16, qkv_bias=True, layer_norm=True, rotary_dim=20)
    else:
        cfg = DL.DenseConfig.tiny(qkv_bias=(kind == "qwen2"))
    W = DL.make_weights(cfg, seed=11)
    common = dict(vocab_size=cfg.vocab, hidden_size=cfg.hidden, intermediate_size=cfg.intermediate,
                  num_hidden_layers=cfg.n_layers, num_attention_heads=cfg.n_heads, num_key_value_heads=cfg.n_kv_heads,
                  max_position_embeddings=cfg.max_seq, tie_word_embeddings=False, attn_implementation="eager")
    rp = {"rope_type": "default", "rope_theta": cfg.rope_theta}
    if kind == "llama":
        hf = tr.LlamaForCausalLM(tr.LlamaConfig(rms_norm_eps=cfg.rms_eps, head_dim=cfg.head_dim, rope_parameters=rp, **common))
    elif kind == "qwen2":
        hf = tr.Qwen2ForCausalLM(tr.Qwen2Config(rms_norm_eps=cfg.rms_eps, rope_parameters=rp, **common))
    else:
        rp["partial_rotary_factor"] = cfg.rotary_dim / cfg.head_dim
        hf = tr.StableLmForCausalLM(tr.StableLmConfig(layer_norm_eps=cfg.rms_eps, use_qkv_bias=True, rope_parameters=rp,
                                                      **common))
    rng = np.random.default_rng(21)
    tokens = [int(t) for t in rng.integers(0, cfg.vocab, 40)]
    want = _hf_logits(hf.float(), _hf_state(W, cfg, cfg.qkv_bias, cfg.layer_norm), tokens)
    orc = DL.OracleDenseLlama(cfg, W, flash_layout=False)
    got = _oracle_prefill_then_decode(orc, cfg, tokens, 3)
    _check_against_hf(got, want, len(tokens) - 4, 3e-2)


def _interleaved_to_half_split(w, n_heads, head_dim):
    """rows of wq / wk: GGUF llama files hold the rows in the order candle's interleaved `rope_i` pairs them
    (quantized_llama.rs:313-318); HF pairs (j, j + D/2).  Same rotation, rows permuted inside each head."""
    w = np.asarray(w).reshape(n_heads, head_dim, -1)
    return np.concatenate([w[:, 0::2], w[:, 1::2]], 1).reshape(n_heads * head_dim, -1)


def test_gguf_oracle_matches_huggingface_llama_on_dequantised_weights():
    """OracleLlama (GGUFLLaMa op order: f32 activations, Q4_K / Q6_K weights, interleaved RoPE, bf16 attention) vs HF
    Llama in f32 on the DEQUANTISED weights with wq / wk rows permuted to HF's pairing."""
    tr = pytest.importorskip("transformers")
    cfg = llama.LlamaConfig.tiny()
    W = llama.make_weights(cfg, seed=4242)

    def deq(tw):
        return kq.dequantize(tw[1], tw[0]).astype(np.float32)
    hfW = {"tok_embd": W["tok_embd"], "output_norm": W["output_norm"], "output": deq(W["output"]), "layers": []}
    for lw in W["layers"]:
        hfW["layers"].append({"wq": _interleaved_to_half_split(deq(lw["wq"]), cfg.n_heads, cfg.head_dim),
                              "wk": _interleaved_to_half_split(deq(lw["wk"]), cfg.n_kv_heads, cfg.head_dim),
                              "wv": deq(lw["wv"]), "wo": deq(lw["wo"]), "w1": deq(lw["w1"]), "w2": deq(lw["w2"]),
                              "w3": deq(lw["w3"]), "attn_norm": lw["attn_norm"], "ffn_norm": lw["ffn_norm"]})
    hf = tr.LlamaForCausalLM(tr.LlamaConfig(
        vocab_size=cfg.vocab, hidden_size=cfg.hidden, intermediate_size=cfg.intermediate, num_hidden_layers=cfg.n_layers,
        num_attention_heads=cfg.n_heads, num_key_value_heads=cfg.n_kv_heads, head_dim=cfg.head_dim,
        max_position_embeddings=cfg.max_seq, rms_norm_eps=cfg.rms_eps, tie_word_embeddings=False,
        rope_parameters={"rope_type": "default", "rope_theta": cfg.rope_theta}, attn_implementation="eager"))
    rng = np.random.default_rng(5)
    tokens = [int(t) for t in rng.integers(0, cfg.vocab, 37)]
    want = _hf_logits(hf.float(), _hf_state(hfW, cfg, False, False), tokens)
    orc = llama.OracleLlama(cfg, W, flash_layout=False)
    got = _oracle_prefill_then_decode(orc, cfg, tokens, 3)
    _check_against_hf(got, want, len(tokens) - 4, 1e-2)


def test_gguf_moe_oracle_matches_huggingface_mixtral():
    """MlpOrMoe::forward restated (softmax -> top-k -> renormalise -> sum_j w_j expert_j(x), quantized_llama.rs:56-123)
    vs HF Mixtral's router + experts in f32 on the dequantised weights (router_top_value /= sum, same top-k)."""
    tr = pytest.importorskip("transformers")
    cfg = llama.LlamaConfig.tiny()
    cfg.n_expert, cfg.n_expert_used = 4, 2
    W = llama.make_moe_weights(cfg, 4, seed=909)

    def deq(tw):
        return kq.dequantize(tw[1], tw[0]).astype(np.float32)
    hf = tr.MixtralForCausalLM(tr.MixtralConfig(
        vocab_size=cfg.vocab, hidden_size=cfg.hidden, intermediate_size=cfg.intermediate, num_hidden_layers=cfg.n_layers,
        num_attention_heads=cfg.n_heads, num_key_value_heads=cfg.n_kv_heads, head_dim=cfg.head_dim,
        max_position_embeddings=cfg.max_seq, rms_norm_eps=cfg.rms_eps, tie_word_embeddings=False, num_local_experts=4,
        num_experts_per_tok=2, sliding_window=None, rope_parameters={"rope_type": "default", "rope_theta": cfg.rope_theta},
        attn_implementation="eager"))
    sd = {"model.embed_tokens.weight": W["tok_embd"], "model.norm.weight": W["output_norm"], "lm_head.weight": deq(W["output"])}
    for l, lw in enumerate(W["layers"]):
        p = f"model.layers.{l}."
        sd[p + "self_attn.q_proj.weight"] = _interleaved_to_half_split(deq(lw["wq"]), cfg.n_heads, cfg.head_dim)
        sd[p + "self_attn.k_proj.weight"] = _interleaved_to_half_split(deq(lw["wk"]), cfg.n_kv_heads, cfg.head_dim)
        sd[p + "self_attn.v_proj.weight"], sd[p + "self_attn.o_proj.weight"] = deq(lw["wv"]), deq(lw["wo"])
        sd[p + "input_layernorm.weight"], sd[p + "post_attention_layernorm.weight"] = lw["attn_norm"], lw["ffn_norm"]
        sd[p + "mlp.gate.weight"] = lw["gate_inp"]
        sd[p + "mlp.experts.gate_up_proj"] = np.stack([np.concatenate([deq(e["w1"]), deq(e["w3"])], 0) for e in lw["experts"]])
        sd[p + "mlp.experts.down_proj"] = np.stack([deq(e["w2"]) for e in lw["experts"]])
    sd = {k: torch.from_numpy(np.ascontiguousarray(v, np.float32)) for k, v in sd.items()}
    rng = np.random.default_rng(6)
    tokens = [int(t) for t in rng.integers(0, cfg.vocab, 29)]
    want = _hf_logits(hf.float(), sd, tokens)
    orc = llama.OracleLlama(cfg, W, flash_layout=False)
    got = _oracle_prefill_then_decode(orc, cfg, tokens, 3)
    _check_against_hf(got, want, len(tokens) - 4, 1e-2)


@pytest.mark.parametrize("scaling", [
    {"rope_type": "llama3", "factor": 8.0, "low_freq_factor": 1.0, "high_freq_factor": 4.0, "original_max_position_embeddings": 16},
    {"rope_type": "yarn", "factor": 4.0, "original_max_position_embeddings": 16, "beta_fast": 32.0, "beta_slow": 1.0},
    {"rope_type": "linear", "factor": 4.0},
], ids=["llama3", "yarn", "linear"])
def test_rope_scaling_through_the_whole_model_matches_huggingface(scaling):
    """`rope_scaling` (llama3 as in Llama-3.1, yarn with its attention factor on cos / sin, linear) applied by the oracle's
    tables vs HF Llama with the same `rope_parameters`; a short original context (16) makes the scaling bite at these
    positions -- without it the two differ by O(1), which the last assert shows"""
    tr = pytest.importorskip("transformers")
    cfg = DL.DenseConfig.tiny()
    W = DL.make_weights(cfg, seed=19)
    rp = dict(scaling)
    rp["rope_theta"] = cfg.rope_theta
    hf = tr.LlamaForCausalLM(tr.LlamaConfig(
        vocab_size=cfg.vocab, hidden_size=cfg.hidden, intermediate_size=cfg.intermediate, num_hidden_layers=cfg.n_layers,
        num_attention_heads=cfg.n_heads, num_key_value_heads=cfg.n_kv_heads, head_dim=cfg.head_dim,
        max_position_embeddings=cfg.max_seq, rms_norm_eps=cfg.rms_eps, tie_word_embeddings=False, rope_parameters=rp,
        attn_implementation="eager"))
    rng = np.random.default_rng(23)
    tokens = [int(t) for t in rng.integers(0, cfg.vocab, 48)]
    want = _hf_logits(hf.float(), _hf_state(W, cfg, False, False), tokens)
    scaled = DL.OracleDenseLlama(cfg, W, flash_layout=False, rope_scaling=scaling, max_position_embeddings=cfg.max_seq)
    got = _oracle_prefill_then_decode(scaled, cfg, tokens, 3)
    _check_against_hf(got, want, len(tokens) - 4, 3e-2)
    plain = _oracle_prefill_then_decode(DL.OracleDenseLlama(cfg, W, flash_layout=False), cfg, tokens, 3)
    assert np.abs(plain - want[len(tokens) - 4:]).max() / np.abs(want).max() > 5e-2     # the scaling really matters here
