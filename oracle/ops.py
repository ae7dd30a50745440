"""CPU oracle (TEST INFRASTRUCTURE ONLY) -- the decode hot path's operators, restated in numpy.

PARITY UNPINNED for the floating-point operators: the reference has no tests, golden vectors or
CPU implementation for them; each function cites the reference
call site whose semantics it follows.  Integer / index operators are exact restatements.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import numpy as np

PAD_SLOT_ID = -1          # src/openai/pipelines/llm_engine.rs:94 (_PAD_SLOT_ID)


# --------------------------------------------------------------------------- bf16 helpers
def f32_to_bf16_bits(x):
    """round-to-nearest-even f32 -> bf16 bit pattern (uint16); NaN preserved as quiet NaN.
    candle `to_dtype(BF16)` = round to nearest even [EXT half crate]; cast sites src/openai/models/layers/attention.rs:977-981."""
    u = np.ascontiguousarray(x, np.float32).view(np.uint32)
    rounding = np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))
    r = ((u + rounding) >> np.uint32(16)).astype(np.uint16)
    nan = np.isnan(np.asarray(x, np.float32))
    if np.any(nan):
        r = np.where(nan, np.uint16(0x7FC0), r)
    return r


def bf16_bits_to_f32(b):
    """inverse of the bf16 cast at src/openai/models/layers/attention.rs:977-981 (exact)."""
    return (np.ascontiguousarray(b, np.uint16).astype(np.uint32) << np.uint32(16)).view(np.float32)


def round_bf16(x):
    """the value a bf16 store keeps: src/openai/models/layers/attention.rs:977-981."""
    return bf16_bits_to_f32(f32_to_bf16_bits(x))


# --------------------------------------------------------------------------- norms / activations
def rms_norm(x, w, eps):
    """candle_nn::ops::rms_norm(x, w, eps): x*rsqrt(mean(x^2)+eps)*w in f32.
    Reference call: src/openai/models/layers/qrmsnorm.rs:28-31."""
    x64 = np.asarray(x, np.float64)
    ms = (x64 * x64).mean(-1, keepdims=True)
    return (x64 / np.sqrt(ms + eps) * np.asarray(w, np.float64)).astype(np.float32)


def silu_mul(gate, up):
    """candle_nn::ops::silu(&w1)? * w3  -- src/openai/models/quantized_llama.rs:33-37."""
    g = np.asarray(gate, np.float64)
    return (g / (1.0 + np.exp(-g)) * np.asarray(up, np.float64)).astype(np.float32)


# --------------------------------------------------------------------------- RoPE
def rope_tables(rope_theta, rotary_dim, max_seq_len):
    """cos/sin [max_seq, rotary_dim/2] f32 -- src/openai/models/layers/rotary_emb.rs:14-48.
    inv_freq[i] = 1f32 / (theta^(2i/dim) as f32) ; idx_theta = pos (f32) * inv_freq (f32 matmul)."""
    inv_freq = _default_inv_freq(rope_theta, rotary_dim)          # 1f32 / (pow(...) as f32): the reciprocal is an f32 op
    pos = np.arange(max_seq_len, dtype=np.float32)[:, None]
    idx_theta = (pos * inv_freq[None, :]).astype(np.float32)
    return np.cos(idx_theta).astype(np.float32), np.sin(idx_theta).astype(np.float32)


def _default_inv_freq(base, dim):
    """calculate_default_inv_freq (rotary_emb.rs:14-19): base^(i/dim) in f64, reciprocal in f32"""
    i = np.arange(0, dim, 2, dtype=np.float64)
    return (np.float32(1.0) / np.power(float(base), i / dim).astype(np.float32)).astype(np.float32)


def rope_tables_scaled(rope_theta, rotary_dim, max_seq_len, scaling=None, max_position_embeddings=0):
    """ScalingRotaryEmbedding::new (rotary_emb.rs:107-341, yarn :358-457): cos/sin f32 [n, rotary_dim/2].
    scaling: None or dict with rope_type in {default, linear, llama3, dynamic, yarn} and the reference's keys."""
    f32 = np.float32
    sc = dict(scaling or {})
    rtype = sc.get("rope_type", sc.get("type", "default"))
    if scaling is None or rtype == "default":
        return rope_tables(rope_theta, rotary_dim, max_seq_len)
    if "original_max_position_embeddings" in sc:
        orig = float(sc["original_max_position_embeddings"])
    elif "factor" in sc and max_position_embeddings:
        orig = float(max_position_embeddings) / float(sc["factor"])
    else:
        orig = float(max_position_embeddings or max_seq_len)

    def tables(inv_freq, n, pos_div=None, mscale=None):
        t = np.arange(n, dtype=f32)
        if pos_div is not None:
            t = (t.astype(np.float64) / float(pos_div)).astype(f32)          # f32 tensor / f64 scalar
        th = (t[:, None] * inv_freq[None, :]).astype(f32)
        c, s = np.cos(th).astype(f32), np.sin(th).astype(f32)
        if mscale is not None:
            c, s = (c.astype(np.float64) * mscale).astype(f32), (s.astype(np.float64) * mscale).astype(f32)
        return c, s

    if rtype == "linear":                                                       # :138-167
        factor = float(sc["factor"])
        return tables(_default_inv_freq(rope_theta, rotary_dim), int(orig * factor), pos_div=factor)
    if rtype == "llama3":                                                       # :168-224
        factor, lo, hi = f32(sc["factor"]), float(sc["low_freq_factor"]), float(sc["high_freq_factor"])
        low_wl, high_wl = f32(orig / lo), f32(orig / hi)
        out = []
        for freq in _default_inv_freq(rope_theta, rotary_dim):
            wavelen = f32(2.0) * f32(np.pi) / freq
            if wavelen < high_wl:
                out.append(freq)
            elif wavelen > low_wl:
                out.append(f32(freq / factor))
            else:
                smooth = f32((f32(orig) / wavelen - f32(lo)) / f32(hi - lo))
                out.append(f32(f32((f32(1.0) - smooth) * freq) / factor + smooth * freq))
        return tables(np.asarray(out, f32), max_seq_len)
    if rtype == "dynamic":                                                      # :227-277
        if "alpha" in sc:
            s_ = float(sc["alpha"])
            n = int(max_position_embeddings)
            theta = (rope_theta * s_) ** (rotary_dim / (rotary_dim - 2))
        else:
            s_ = float(sc["factor"])
            n = int(orig * s_)
            theta = (rope_theta * ((s_ * n / orig) - (s_ - 1.0))) ** (rotary_dim / (rotary_dim - 2))
        return tables(_default_inv_freq(theta, rotary_dim), n)
    if rtype == "yarn":                                                         # :278-320, :400-457
        factor = f32(sc["factor"])
        beta_fast, beta_slow = f32(sc.get("beta_fast", 32.0)), f32(sc.get("beta_slow", 1.0))
        attn_factor, extrap = f32(sc.get("attn_factor", 1.0)), f32(sc.get("extrapolation_factor", 1.0))
        base, dim = f32(rope_theta), rotary_dim

        def corr(num_rot):
            return f32(f32(dim) * np.log(f32(f32(int(orig)) / f32(num_rot * f32(2.0) * f32(np.pi))))) / f32(f32(2.0) * np.log(base))
        low, high = max(np.floor(corr(beta_fast)), f32(0.0)), min(np.ceil(corr(beta_slow)), f32(dim - 1))
        if low == high:
            high = f32(high + f32(0.001))
        k = np.arange(dim // 2, dtype=f32)
        pw = np.power(base, (2 * k / f32(dim)).astype(f32)).astype(f32)
        extra, inter = (f32(1.0) / pw).astype(f32), (f32(1.0) / (factor * pw).astype(f32)).astype(f32)
        ramp = ((k.astype(np.float64) - float(low)).astype(f32).astype(np.float64) / (float(high) - float(low))).astype(f32)
        ramp = np.clip(ramp, f32(0.0), f32(1.0))
        mask = ((1.0 - ramp.astype(np.float64)).astype(f32).astype(np.float64) * float(extrap)).astype(f32)
        inv = (inter * (1.0 - mask.astype(np.float64)).astype(f32) + extra * mask).astype(f32)
        mp = int(max_position_embeddings or max_seq_len)
        mscale = (f32(1.0) if factor <= 1 else f32(f32(0.1) * np.log(factor) + f32(1.0))) * attn_factor
        return tables(inv, int(f32(mp) * factor), mscale=float(mscale))
    raise ValueError(f"Unknown rope_type: {rtype}")


def rope_apply(x, cos, sin, positions, interleaved, rotary_dim=None):
    """x [T, H, D] f32; returns rotated copy.

    interleaved=True  -> candle `rope_i`: pairs (x[2i], x[2i+1])      (GGUF llama,
                         src/openai/models/quantized_llama.rs:313-318 passes is_gpt_neox=false)
    interleaved=False -> candle `rope`  : pairs (x[i], x[i+rot/2])    (HF llama, llama.rs:222)
    Partial rotary: only the first `rotary_dim` channels rotate (rotary_emb.rs:80-94).
    """
    x = np.asarray(x, np.float32)
    T, H, D = x.shape
    rot = D if rotary_dim is None else rotary_dim
    c = cos[np.asarray(positions)][:, None, : rot // 2].astype(np.float32)
    s = sin[np.asarray(positions)][:, None, : rot // 2].astype(np.float32)
    out = x.copy()
    xr = x[..., :rot]
    if interleaved:
        x0 = xr[..., 0::2]
        x1 = xr[..., 1::2]
        out[..., 0:rot:2] = x0 * c - x1 * s
        out[..., 1:rot:2] = x0 * s + x1 * c
    else:
        x0 = xr[..., : rot // 2]
        x1 = xr[..., rot // 2:]
        out[..., : rot // 2] = x0 * c - x1 * s
        out[..., rot // 2: rot] = x0 * s + x1 * c
    return out


# --------------------------------------------------------------------------- KV cache layouts
def kv_cache_shapes(num_blocks, block_size, num_kv_heads, head_dim, elem_size, flash_layout):
    """src/scheduler/cache_engine.rs:298-341.  Returns (key_shape, value_shape)."""
    if flash_layout:
        s = (num_blocks, block_size, num_kv_heads, head_dim)
        return s, s
    x = 16 // elem_size
    return ((num_blocks, num_kv_heads, head_dim // x, block_size, x),
            (num_blocks, num_kv_heads, head_dim, block_size))


def reshape_and_cache(k, v, key_cache, value_cache, slot_mapping, flash_layout):
    """Scatter new K/V rows into the paged cache (in place).  k, v: [T, Hkv, D] (any dtype array,
    copied bit-exactly); slot = block*block_size + offset; slot < 0 -> skipped.
    Reference call: PagedAttention::forward, src/openai/models/layers/attention.rs:983-995;
    slot construction src/openai/pipelines/inputs.rs:410-423."""
    T, Hkv, D = k.shape
    if flash_layout:
        bs = key_cache.shape[1]
        for t in range(T):
            slot = int(slot_mapping[t])
            if slot < 0:
                continue
            key_cache[slot // bs, slot % bs] = k[t]
            value_cache[slot // bs, slot % bs] = v[t]
    else:
        bs = key_cache.shape[3]
        x = key_cache.shape[4]
        for t in range(T):
            slot = int(slot_mapping[t])
            if slot < 0:
                continue
            blk, off = slot // bs, slot % bs
            key_cache[blk, :, :, off, :] = k[t].reshape(Hkv, D // x, x)
            value_cache[blk, :, :, off] = v[t]


def gather_kv(key_cache, value_cache, block_table, context_len, flash_layout):
    """Collect [context_len, Hkv, D] K and V for one sequence through its block table.
    reads the cache layouts of src/scheduler/cache_engine.rs:298-341 through one sequence's block table (pipelines/inputs.rs:425-454)."""
    if flash_layout:
        bs = key_cache.shape[1]
    else:
        bs = key_cache.shape[3]
    ks, vs = [], []
    for pos in range(context_len):
        blk = int(block_table[pos // bs])
        off = pos % bs
        if flash_layout:
            ks.append(key_cache[blk, off])
            vs.append(value_cache[blk, off])
        else:
            kk = key_cache[blk, :, :, off, :]
            ks.append(kk.reshape(kk.shape[0], -1))
            vs.append(value_cache[blk, :, :, off])
    return np.stack(ks), np.stack(vs)


def paged_attention_decode(q, key_cache, value_cache, block_tables, context_lens, scale,
                           flash_layout, softcap=None, kv_is_bf16_bits=True, sliding_window=None):
    """Decode attention over the paged cache.  q: [B, H, D] f32 values (already bf16-rounded by the
    caller, as attention.rs:977-981 casts q,k,v to bf16).  Caches hold bf16 bit patterns (uint16)
    when kv_is_bf16_bits else float arrays.  Math = NaiveAttention, src/openai/models/mod.rs:1288-1306
    (repeat_kv GQA expansion :1240-1247, q.k^T*scale, optional tanh softcap, softmax_last_dim, .v),
    accumulated in f64; output rounded to bf16 (the kernel's output dtype).  Returns f32 [B, H, D].
    sliding_window (`PagedAttention::new(.., sliding_window, ..)`, attention.rs:566-575,888-897; the kernel lives in the un-vendored
    attention-rs [EXT]): the query at position n - 1 sees the last `sliding_window` keys only, positions n - w .. n - 1 -- the published
    semantics of the models that carry the field (Mistral / HF `sliding_window`: key j visible to query i iff i - j < w) and of vLLM's
    paged attention, which truncates the block table to the window."""
    B, H, D = q.shape
    out = np.zeros((B, H, D), np.float32)
    for b in range(B):
        n = int(context_lens[b])
        if n == 0:
            continue
        k, v = gather_kv(key_cache, value_cache, block_tables[b], n, flash_layout)
        if kv_is_bf16_bits:
            k = bf16_bits_to_f32(k)
            v = bf16_bits_to_f32(v)
        if sliding_window is not None and sliding_window > 0 and n > sliding_window:
            k, v = k[n - sliding_window:], v[n - sliding_window:]
        Hkv = k.shape[1]
        g = H // Hkv
        for h in range(H):
            kh = k[:, h // g, :].astype(np.float64)
            vh = v[:, h // g, :].astype(np.float64)
            s = kh @ q[b, h].astype(np.float64) * scale
            if softcap is not None:
                s = np.tanh(s / softcap) * softcap
            s = s - s.max()
            p = np.exp(s)
            p /= p.sum()
            out[b, h] = (p @ vh).astype(np.float32)
    return round_bf16(out)


def paged_attention_decode_bf16_tensors(q, key_cache, value_cache, block_tables, context_lens, scale, flash_layout):
    """The same attention with the reference CPU path's OWN rounding points (NaiveAttention::forward on bf16 tensors,
    src/openai/models/mod.rs:1288-1306): `q.matmul(k^T)` returns bf16, `* scale` rounds to bf16 again, `softmax_last_dim`
    returns bf16 probabilities (computed in f32 from the bf16 scores), `attn.matmul(v)` accumulates in f32 and returns bf16.
    Summation in f32, in index order (what oracle.c's bf16-attention mode and the GPU's parity-mode kernel do)."""
    B, H, D = q.shape
    out = np.zeros((B, H, D), np.float32)
    for b in range(B):
        n = int(context_lens[b])
        if n == 0:
            continue
        k, v = gather_kv(key_cache, value_cache, block_tables[b], n, flash_layout)
        k = bf16_bits_to_f32(k)
        v = bf16_bits_to_f32(v)
        g = H // k.shape[1]
        for h in range(H):
            kh, vh, qh = k[:, h // g, :], v[:, h // g, :], q[b, h].astype(np.float32)
            s = np.zeros(n, np.float32)
            for d in range(D):                                         # index order, f32
                s = s + kh[:, d] * qh[d]
            s = round_bf16(round_bf16(s) * np.float32(scale))
            e = np.exp((s - s.max()).astype(np.float32)).astype(np.float32)
            p = round_bf16((e.astype(np.float64) / e.astype(np.float64).sum()).astype(np.float32))
            o = np.zeros(D, np.float32)
            for t in range(n):
                o = o + p[t] * vh[t]
            out[b, h] = round_bf16(o)
    return out


def copy_blocks(key_caches, value_caches, block_mapping_pairs):
    """src/backend/cache.rs:103-162: for every layer and every (src,dst) pair copy block src->dst
    of K and of V (dim 0 = block).  block_mapping_pairs: flat [src0,dst0,src1,dst1,...]."""
    pairs = np.asarray(block_mapping_pairs, np.int64).reshape(-1, 2)
    for kc, vc in zip(key_caches, value_caches):
        for src, dst in pairs:
            kc[dst] = kc[src]
            vc[dst] = vc[src]


def swap_blocks(src, dst, mapping):
    """attention_rs::cache::swap_blocks(src, dst, &HashMap<src_blk,dst_blk>) --
    src/scheduler/cache_engine.rs:527-535: per-block copy between two tensors whose dim 0 = block."""
    for s, d in mapping.items():
        dst[d] = src[s]


# --------------------------------------------------------------------------- input preparation (integer, bit-exact)
def used_blocks_for_len(seq_len, block_size, table_len):
    """src/openai/pipelines/inputs.rs:12-22."""
    if seq_len == 0:
        return 0
    return min(-(-seq_len // block_size), table_len)


def prepare_decode(seqs, block_size):
    """src/openai/pipelines/inputs.rs:376-454.  seqs: list of dicts
    {"tokens": [...all token ids so far...], "block_table": [block ids]} in batch order.
    Returns dict of int arrays exactly as the reference builds them."""
    input_ids, positions, slot_mapping, context_lens, tables = [], [], [], [], []
    for s in seqs:
        toks = s["tokens"]
        table = list(s["block_table"])
        n = len(toks)
        input_ids.append(toks[-1])
        position = n - 1
        positions.append(position)
        context_lens.append(n)
        if position // block_size >= len(table):
            raise ValueError("Block table is too small (completion)!")
        slot_mapping.append(table[position // block_size] * block_size + position % block_size)
        tables.append(table[: used_blocks_for_len(n, block_size, len(table))])
    maxlen = max(len(t) for t in tables)
    bt = np.zeros((len(tables), maxlen), np.uint32)       # _make_tensor_with_pad(..., pad=0)
    for i, t in enumerate(tables):
        bt[i, : len(t)] = t
    return {
        "input_ids": np.asarray(input_ids, np.uint32),
        "positions": np.asarray(positions, np.int64),
        "slot_mapping": np.asarray(slot_mapping, np.int64),
        "context_lens": np.asarray(context_lens, np.uint32),
        "block_tables": bt,
        "max_context_len": int(max(context_lens)),
    }


def prepare_prompt(seqs, block_size, num_cached_tokens=None):
    """src/openai/pipelines/inputs.rs:90-230 (non-flashinfer part): flatten prompts, slot per
    token, cu_seqlens_q/k, context_lens = cached + chunk.  seqs: {"tokens", "block_table"};
    num_cached_tokens[i] tokens of sequence i are already in the cache (prefix cache / chunk)."""
    input_ids, positions, slot_mapping = [], [], []
    cu_q, cu_k, context_lens = [0], [0], []
    tables = []
    for i, s in enumerate(seqs):
        cached = 0 if num_cached_tokens is None else int(num_cached_tokens[i])
        toks = s["tokens"][cached:]
        table = list(s["block_table"])
        for j, t in enumerate(toks):
            pos = cached + j
            input_ids.append(t)
            positions.append(pos)
            if pos // block_size >= len(table):
                slot_mapping.append(PAD_SLOT_ID)
            else:
                slot_mapping.append(table[pos // block_size] * block_size + pos % block_size)
        cu_q.append(cu_q[-1] + len(toks))
        cu_k.append(cu_k[-1] + cached + len(toks))
        context_lens.append(cached + len(toks))
        tables.append(table)
    maxlen = max(len(t) for t in tables)
    bt = np.zeros((len(tables), maxlen), np.uint32)
    for i, t in enumerate(tables):
        bt[i, : len(t)] = t
    return {
        "input_ids": np.asarray(input_ids, np.uint32),
        "positions": np.asarray(positions, np.int64),
        "slot_mapping": np.asarray(slot_mapping, np.int64),
        "context_lens": np.asarray(context_lens, np.uint32),
        "block_tables": bt,
        "cu_seqlens_q": np.asarray(cu_q, np.uint32),
        "cu_seqlens_k": np.asarray(cu_k, np.uint32),
        "max_seqlen_q": int(max(np.diff(cu_q))),
        "max_seqlen_k": int(max(np.diff(cu_k))),
        "max_context_len": int(max(context_lens)),
    }


def num_gpu_blocks(mem_mb, dsize, block_size, num_kv_heads_local, head_dim, num_layers):
    """get_cache_config: src/lib.rs:181-188 -- blocks = mem_MB*2^20 / (dsize*bs*Hkv*D*layers*2)."""
    return (mem_mb * 1024 * 1024) // (dsize * block_size * num_kv_heads_local * head_dim * num_layers * 2)


def kv_head_shard(total_kv_heads, rank, world_size):
    """src/openai/distributed.rs:725-765 -> (local_kv_heads, shard_rank, shard_world)."""
    if total_kv_heads == 0 or world_size == 0 or rank >= world_size:
        raise ValueError("bad tensor-parallel arguments")
    if total_kv_heads >= world_size:
        if total_kv_heads % world_size:
            raise ValueError("KV heads must be divisible by world_size")
        return total_kv_heads // world_size, rank, world_size
    if world_size % total_kv_heads:
        raise ValueError("world_size must be divisible by KV heads")
    return 1, rank // (world_size // total_kv_heads), total_kv_heads


def prefill_attention(q, k, v, scale, softcap=None, cached=0, rnd=None, sliding_window=None):
    """Causal self-attention for ONE sequence.  q [T,H,D] (the chunk), k/v [cached+T,Hkv,D] (cached prefix
    followed by the chunk; `use_cached_kv`, inputs.rs:133-143) -- f32 values already rounded to the 16-bit dtype.
    NaiveAttention math (models/mod.rs:1288-1306) with the causal additive mask of layers/mask.rs:32-53:
    query t sees keys 0 .. cached+t.  Output rounded by `rnd` (default bf16) f32 [T,H,D].
    sliding_window (layers/mask.rs:22-27 hands it to attention-rs' `causal_mask` [EXT]): the query at position i = cached + t sees keys
    max(0, i - w + 1) .. i (HF: masked iff kv_idx <= q_idx - w)."""
    T, H, D = q.shape
    Hkv = k.shape[1]
    g = H // Hkv
    out = np.zeros((T, H, D), np.float32)
    mask = np.triu(np.ones((T, cached + T), bool), cached + 1)
    if sliding_window is not None and sliding_window > 0:
        qi = cached + np.arange(T)[:, None]
        mask = mask | (np.arange(cached + T)[None, :] <= qi - sliding_window)
    for h in range(H):
        s = q[:, h].astype(np.float64) @ k[:, h // g].astype(np.float64).T * scale
        if softcap is not None:
            s = np.tanh(s / softcap) * softcap
        s = np.where(mask, -np.inf, s)
        s = s - s.max(-1, keepdims=True)
        p = np.exp(s)
        p /= p.sum(-1, keepdims=True)
        out[:, h] = (p @ v[:, h // g].astype(np.float64)).astype(np.float32)
    return round_bf16(out) if rnd is None else rnd(out)


# ------------------------------------------------------------------------------------------------ fp8 KV cache
# `--kvcache-dtype fp8`: the cache tensors are U8 (src/main.rs:263-267), K layout x = 16 for 1-byte elements
# (cache_engine.rs:304-311); PagedAttention is built with is_fp8_keys (attention.rs:574,896).  The conversion kernels
# live in attention-rs (un-vendored): OCP e4m3fn, round-to-nearest-even, saturating at +-448, scale 1.0 [EXT].
def f32_to_e4m3fn(x):
    """f32 -> OCP e4m3fn bytes (uint8): RNE, saturate to +-448 (0x7E), NaN -> 0x7F."""
    x = np.asarray(x, np.float32)
    isnan = np.isnan(x)
    sign = (np.signbit(x)).astype(np.uint8) << 7
    a = np.minimum(np.abs(np.where(isnan, np.float32(0), x).astype(np.float64)), 448.0)   # NaN lanes are overwritten below
    out = np.zeros(x.shape, np.uint8)
    nz = a > 0
    e = np.floor(np.log2(a, where=nz, out=np.zeros_like(a)))
    e = np.clip(e, -6, 8)                                  # subnormals share exponent -6
    step = np.exp2(e - 3)                                  # 3 mantissa bits
    q = np.rint(a / step) * step                           # numpy rint = round half to even
    q = np.minimum(q, 448.0)
    e2 = np.floor(np.log2(q, where=q > 0, out=np.zeros_like(q)))
    e2 = np.clip(e2, -6, 8)
    man = np.rint(q / np.exp2(e2 - 3)).astype(np.int64)   # 8..15 normal, 0..7 subnormal
    normal = man >= 8
    bits = np.where(normal, ((e2 + 7).astype(np.int64) << 3) | (man - 8), man)
    out = np.where(nz, bits, 0).astype(np.uint8)
    out = np.where(isnan, np.uint8(0x7F), out | sign)
    return out.astype(np.uint8)


def e4m3fn_to_f32(b):
    """OCP e4m3fn decode for the U8 cache of src/main.rs:263-267 (cache_engine.rs:304-311)."""
    b = np.asarray(b, np.uint8).astype(np.int64)
    s = np.where(b & 0x80, -1.0, 1.0)
    e = (b >> 3) & 0xF
    m = b & 7
    v = np.where(e == 0, m / 8.0 * 2.0 ** -6, (1.0 + m / 8.0) * np.exp2(e - 7.0))
    v = np.where((e == 15) & (m == 7), np.nan, v)
    return (s * v).astype(np.float32)


def reshape_and_cache_fp8(k_f32, v_f32, key_cache_u8, value_cache_u8, slot_mapping, flash_layout=False):
    """k, v values [T,Hkv,D] (already rounded to the model dtype) -> e4m3fn bytes scattered like reshape_and_cache.
    K1 on the U8 cache: src/scheduler/cache_engine.rs:304-311, call site src/openai/models/layers/attention.rs:983-995 (is_fp8_keys :896)."""
    reshape_and_cache(f32_to_e4m3fn(k_f32), f32_to_e4m3fn(v_f32), key_cache_u8, value_cache_u8, slot_mapping, flash_layout)


def fp8_cache_as_bf16_bits(key_cache_u8, value_cache_u8, flash_layout=False):
    """e4m3fn caches -> bf16-bit caches in the 2-byte layouts (every e4m3 value is exact in bf16), so the 16-bit
    attention restatements apply unchanged.  PAGED K: [NB,Hkv,D/16,bs,16] -> [NB,Hkv,D/8,bs,8].
    what the fp8 attention kernels read back: src/openai/models/layers/attention.rs:574,896."""
    kf = f32_to_bf16_bits(e4m3fn_to_f32(key_cache_u8))
    vf = f32_to_bf16_bits(e4m3fn_to_f32(value_cache_u8))
    if not flash_layout:
        nb, h, d16, s, x = kf.shape
        kf = kf.reshape(nb, h, d16, s, 2, 8).transpose(0, 1, 2, 4, 3, 5).reshape(nb, h, d16 * 2, s, 8)
    return np.ascontiguousarray(kf), np.ascontiguousarray(vf)
