"""CPU suite (-m "not gpu"): the oracle against its own invariants / committed golden vectors, the host
logic, the layout arithmetic of the MFMA kernel (lane-level emulation on the real repack output) and the
C-ABI export check.  No compute call touches a GPU here."""
import ctypes
import json
import os

import numpy as np
import pytest

from oracle import kquants as kq
from oracle import llama
from oracle import ops as O

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


# ------------------------------------------------------------------------------------------------ k-quants
def test_q4k_scale_pack_roundtrip_all_values():
    sc = np.arange(64, dtype=np.uint8).reshape(8, 8)
    m = sc[::-1].copy()
    s2, m2 = kq._unpack_scales_q4k(kq._pack_scales_q4k(sc, m))
    assert np.array_equal(s2, sc) and np.array_equal(m2, m)


def test_q4k_known_answer_block():
    """Hand-built block: d=1, dmin=0.5, sc_j=j+1, m_j=2j, qs byte = 0x21 -> lo nibble 1, hi nibble 2."""
    blk = np.zeros(144, np.uint8)
    blk[0:2] = np.array([1.0], np.float16).view(np.uint8)
    blk[2:4] = np.array([0.5], np.float16).view(np.uint8)
    sc = np.arange(1, 9, dtype=np.uint8)[None]
    mn = (2 * np.arange(8, dtype=np.uint8))[None]
    blk[4:16] = kq._pack_scales_q4k(sc, mn)[0]
    blk[16:] = 0x21
    w = kq.dequantize_q4_k(blk[None, None])[0]
    for j in range(8):
        q = 1 if j % 2 == 0 else 2
        assert np.allclose(w[32 * j:32 * j + 32], (j + 1) * q - 0.5 * (2 * j))


def test_q6k_known_answer_block():
    """ql=0x4A (lo 10, hi 4), qh=0b11100100 -> q1..q4 = 10|0<<4, 10|1<<4, 4|2<<4, 4|3<<4; scales = 1..16, d=2."""
    blk = np.zeros(210, np.uint8)
    blk[0:128] = 0x4A
    blk[128:192] = 0b11100100
    blk[192:208] = np.arange(1, 17, dtype=np.int8).view(np.uint8)
    blk[208:210] = np.array([2.0], np.float16).view(np.uint8)
    w = kq.dequantize_q6_k(blk[None, None])[0]
    qv = [10, 26, 36, 52]
    for n in range(2):
        for t in range(4):
            for half in range(2):
                sc = 8 * n + 2 * t + half + 1
                seg = w[128 * n + 32 * t + 16 * half: 128 * n + 32 * t + 16 * half + 16]
                assert np.allclose(seg, 2.0 * sc * (qv[t] - 32))


@pytest.mark.parametrize("t,tol", [(kq.GGML_Q4_K, 0.08), (kq.GGML_Q6_K, 0.04), (kq.GGML_Q8_0, 0.006)])
def test_quantize_dequantize_roundtrip(t, tol):
    rng = np.random.default_rng(0)
    w = rng.normal(0, 0.02, (16, 1024)).astype(np.float32)
    d = kq.dequantize(kq.quantize(w, t), t)
    assert np.abs(d - w).max() < tol * np.abs(w).max()
    # idempotence: re-quantising dequantised weights reproduces them (up to f16 scale rounding)
    d2 = kq.dequantize(kq.quantize(d, t), t)
    assert np.abs(d2 - d).max() < 0.35 * tol * np.abs(w).max()


def test_o2_close_to_o1():
    """candle-CPU-faithful Q8_K path differs from exact dequant arithmetic by ~1 %."""
    rng = np.random.default_rng(1)
    x = rng.normal(size=(2, 2048)).astype(np.float32)
    for t in (kq.GGML_Q4_K, kq.GGML_Q6_K):
        b = kq.quantize(rng.normal(0, 0.02, (32, 2048)).astype(np.float32), t)
        o1, o2 = kq.qmatmul_o1(x, b, t), kq.qmatmul_o2(x, b, t)
        err = np.abs(o1 - o2).max() / np.abs(o1).max()
        assert 1e-5 < err < 3e-2


# ------------------------------------------------------------------------------------------------ layout arithmetic
@pytest.mark.parametrize("t", [kq.GGML_Q4_K, kq.GGML_Q6_K])
@pytest.mark.parametrize("B,N,K", [(1, 32, 512), (2, 16, 256), (3, 24, 512), (8, 16, 256)])
def test_mfma_kernel_layout_emulation(lib, t, B, N, K):
    from tests import qmm_emulator as emu
    rng = np.random.default_rng(2)
    blocks = kq.quantize(rng.normal(0, 0.05, (N, K)).astype(np.float32), t)
    x = rng.normal(size=(B, K)).astype(np.float32)
    n = lib.mi355_qweight_repacked_size(t, N, K)
    tiles = np.zeros(n, np.uint8)
    src = np.ascontiguousarray(blocks)
    assert lib.mi355_qweight_repack(tiles.ctypes.data, src.ctypes.data, t, N, K) == 0
    got = emu.emulate_qmatmul(x, tiles, t, N, K)
    ref = kq.qmatmul_o1(x, blocks, t)
    assert np.abs(got - ref).max() < 2e-5 * np.abs(ref).max()


def test_repack_is_a_permutation_of_bytes(lib):
    rng = np.random.default_rng(3)
    for t, bb in ((kq.GGML_Q4_K, 144), (kq.GGML_Q6_K, 210)):
        N, K = 32, 512
        src = rng.integers(0, 256, (N, K // 256, bb)).astype(np.uint8)
        n = lib.mi355_qweight_repacked_size(t, N, K)
        assert n == src.size                                     # same bytes per weight
        dst = np.zeros(n, np.uint8)
        assert lib.mi355_qweight_repack(dst.ctypes.data, src.ctypes.data, t, N, K) == 0
        assert np.array_equal(np.sort(dst), np.sort(src.ravel()))
    assert lib.mi355_qweight_repacked_size(kq.GGML_Q4_K, 16, 100) == -1
    assert lib.mi355_qweight_repacked_size(2, 16, 256) == -1


# ------------------------------------------------------------------------------------------------ integer host logic
def test_prepare_decode_matches_reference_rules():
    """inputs.rs:376-454: position = len-1, slot = table[pos/bs]*bs + pos%bs, tables truncated to used
    blocks and 0-padded."""
    seqs = [{"tokens": list(range(130)), "block_table": [7, 3, 9, 11]},
            {"tokens": list(range(64)), "block_table": [5]},
            {"tokens": [42], "block_table": [2, 8]}]
    m = O.prepare_decode(seqs, 64)
    assert m["positions"].tolist() == [129, 63, 0]
    assert m["slot_mapping"].tolist() == [9 * 64 + 1, 5 * 64 + 63, 2 * 64]
    assert m["context_lens"].tolist() == [130, 64, 1]
    assert m["block_tables"].tolist() == [[7, 3, 9], [5, 0, 0], [2, 0, 0]]
    assert m["input_ids"].tolist() == [129, 63, 42] and m["max_context_len"] == 130
    with pytest.raises(ValueError):
        O.prepare_decode([{"tokens": list(range(65)), "block_table": [1]}], 64)


def test_used_blocks_and_cache_budget():
    assert O.used_blocks_for_len(0, 64, 4) == 0 and O.used_blocks_for_len(64, 64, 4) == 1
    assert O.used_blocks_for_len(65, 64, 4) == 2 and O.used_blocks_for_len(1000, 64, 4) == 4
    # lib.rs:181-188 with llama-3-8B bf16 bs=64: one block (all layers, K+V) = 8 MiB
    assert O.num_gpu_blocks(8 * 1024, 2, 64, 8, 128, 32) == 1024


def test_kv_head_shard():
    assert O.kv_head_shard(8, 3, 8) == (1, 3, 8)
    assert O.kv_head_shard(8, 1, 2) == (4, 1, 2)
    assert O.kv_head_shard(4, 5, 8) == (1, 2, 4)       # replicated: ranks 4,5 share kv head 2
    with pytest.raises(ValueError):
        O.kv_head_shard(6, 0, 4)


def test_cache_layout_index_formulas():
    """SURVEY App. B element index formulas == what reshape_and_cache writes."""
    NB, bs, Hkv, D = 3, 16, 2, 32
    for flash in (True, False):
        ks, vs = O.kv_cache_shapes(NB, bs, Hkv, D, 2, flash)
        kc, vc = np.zeros(ks, np.uint16), np.zeros(vs, np.uint16)
        k = np.arange(1, Hkv * D + 1, dtype=np.uint16).reshape(1, Hkv, D)
        O.reshape_and_cache(k, k + 1000, kc, vc, np.array([bs * 2 + 5]), flash)
        blk, off, h, d, x = 2, 5, 1, 19, 8
        if flash:
            assert kc.ravel()[((blk * bs + off) * Hkv + h) * D + d] == k[0, h, d]
        else:
            assert kc.ravel()[(((blk * Hkv + h) * (D // x) + d // x) * bs + off) * x + d % x] == k[0, h, d]
            assert vc.ravel()[((blk * Hkv + h) * D + d) * bs + off] == k[0, h, d] + 1000


# ------------------------------------------------------------------------------------------------ fp ops & model
def test_rope_styles_agree_under_permutation():
    """interleaved RoPE on llama.cpp-permuted channels == half-split RoPE on the original channels."""
    rng = np.random.default_rng(4)
    T, H, D = 5, 3, 64
    x = rng.normal(size=(T, H, D)).astype(np.float32)
    pos = np.array([0, 1, 7, 100, 255])
    cos, sin = O.rope_tables(10000.0, D, 256)
    neox = O.rope_apply(x, cos, sin, pos, interleaved=False)
    perm = np.empty(D, np.int64)
    perm[0::2] = np.arange(D // 2)
    perm[1::2] = np.arange(D // 2) + D // 2
    inter = O.rope_apply(x[..., perm], cos, sin, pos, interleaved=True)
    assert np.allclose(inter, neox[..., perm], atol=1e-6)


def test_paged_attention_oracle_layout_invariance_and_softmax():
    rng = np.random.default_rng(5)
    B, H, Hkv, D, bs = 2, 4, 2, 32, 16
    ctx = [37, 16]
    outs = []
    kf = rng.normal(size=(60, Hkv, D)).astype(np.float32)
    vf = rng.normal(size=(60, Hkv, D)).astype(np.float32)
    q = O.round_bf16(rng.normal(size=(B, H, D)).astype(np.float32))
    bt = np.array([[2, 0, 1], [3, 0, 0]], np.uint32)
    for flash in (True, False):
        ks, vs = O.kv_cache_shapes(4, bs, Hkv, D, 2, flash)
        kc, vc = np.zeros(ks, np.uint16), np.zeros(vs, np.uint16)
        t = 0
        for b, c in enumerate(ctx):
            slots = np.array([int(bt[b, p // bs]) * bs + p % bs for p in range(c)])
            O.reshape_and_cache(O.f32_to_bf16_bits(kf[t:t + c]), O.f32_to_bf16_bits(vf[t:t + c]), kc, vc, slots, flash)
            t += c
        outs.append(O.paged_attention_decode(q, kc, vc, bt, np.array(ctx), 0.2, flash))
    assert np.array_equal(outs[0], outs[1])
    # direct dense check of one head
    k0 = O.round_bf16(kf[:37, 0]).astype(np.float64)
    v0 = O.round_bf16(vf[:37, 0]).astype(np.float64)
    s = k0 @ q[0, 1].astype(np.float64) * 0.2
    p = np.exp(s - s.max())
    p /= p.sum()
    assert np.allclose(outs[0][0, 1], O.round_bf16((p @ v0).astype(np.float32)))


def test_tiny_llama_decode_equals_prefill_and_golden():
    """Layer wiring: decoding token n through the paged cache == prefilling n+1 tokens; logits pinned by a
    committed fixture (tests/golden/tiny_llama_logits.json, made by tests/golden/make_golden.py)."""
    cfg = llama.LlamaConfig.tiny()
    W = llama.make_weights(cfg, seed=1234)
    M = llama.OracleLlama(cfg, W)
    rng = np.random.default_rng(7)
    seqs = [{"tokens": [int(t) for t in rng.integers(0, cfg.vocab, 19)], "block_table": [3, 7]},
            {"tokens": [int(t) for t in rng.integers(0, cfg.vocab, 5)], "block_table": [1]}]
    cache = M.new_cache(16)
    lg = M.forward(O.prepare_prompt(seqs, cfg.block_size), cache, is_prefill=True)
    for s, row in zip(seqs, lg):
        s["tokens"].append(int(row.argmax()))
    dec = M.forward(O.prepare_decode(seqs, cfg.block_size), cache)
    cache2 = M.new_cache(16)
    full = M.forward(O.prepare_prompt(seqs, cfg.block_size), cache2, is_prefill=True)
    assert np.abs(dec - full).max() < 1e-5 * np.abs(full).max()
    path = os.path.join(GOLDEN, "tiny_llama_logits.json")
    gold = json.load(open(path))
    assert gold["next_tokens"] == [int(r.argmax()) for r in dec]
    assert np.allclose(np.asarray(gold["logits_head"], np.float32), dec[:, :16], rtol=0, atol=2e-5)


# ------------------------------------------------------------------------------------------------ C ABI
def test_c_abi_exports_every_declared_symbol(lib):
    from candle_vllm_amd import _lib
    names = _lib.declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), n
    for n in ("copy_blocks_bf16", "copy_blocks_f16", "copy_blocks_f32"):       # reference FFI, verbatim
        assert n in names


# ------------------------------------------------------------------------------------------------ C twin of the oracle
def test_c_oracle_matches_numpy_oracle():
    from oracle import cref
    cref.build()
    rng = np.random.default_rng(21)
    x = rng.normal(size=(3, 1024)).astype(np.float32)
    for t in (kq.GGML_Q4_K, kq.GGML_Q6_K):
        b = kq.quantize(rng.normal(0, 0.05, (24, 1024)).astype(np.float32), t)
        o1 = kq.qmatmul_o1(x, b, t)
        assert np.abs(cref.qmatmul(x, b, t, o2=False) - o1).max() < 2e-6 * np.abs(o1).max()
        o2 = kq.qmatmul_o2(x, b, t)
        assert np.abs(cref.qmatmul(x, b, t, o2=True) - o2).max() < 2e-5 * np.abs(o2).max()


def test_c_oracle_decode_step_matches_numpy_llama():
    from oracle import cref
    cref.build()
    cfg = llama.LlamaConfig.tiny()
    W = llama.make_weights(cfg, seed=1234)
    M = llama.OracleLlama(cfg, W)
    rng = np.random.default_rng(7)
    seqs = [{"tokens": [int(t) for t in rng.integers(0, cfg.vocab, 19)], "block_table": [3, 7]},
            {"tokens": [int(t) for t in rng.integers(0, cfg.vocab, 5)], "block_table": [1]}]
    cache = M.new_cache(16)
    lg = M.forward(O.prepare_prompt(seqs, cfg.block_size), cache, is_prefill=True)
    for s, row in zip(seqs, lg):
        s["tokens"].append(int(row.argmax()))
    meta = O.prepare_decode(seqs, cfg.block_size)
    c_cache = [(k.copy(), v.copy()) for k, v in cache]
    ref = M.forward(meta, cache)
    got = cref.CLlama(cfg, W).decode(meta, c_cache, o2=False)
    assert np.abs(got - ref).max() < 2e-4 * np.abs(ref).max()
    for (k1, v1), (k2, v2) in zip(cache, c_cache):
        assert np.abs(O.bf16_bits_to_f32(k1) - O.bf16_bits_to_f32(k2)).max() <= 2 ** -7 * np.abs(O.bf16_bits_to_f32(k1)).max()
    # candle-CPU-faithful arithmetic (Q8_K activations) stays within ~a few % of the exact-dequant logits
    got2 = cref.CLlama(cfg, W).decode(meta, [(k.copy(), v.copy()) for k, v in c_cache], o2=True)
    assert np.abs(got2 - ref).max() < 0.1 * np.abs(ref).max()


def test_c_oracle_o1f_blocked_products_and_prompt_step_match_numpy_llama():
    """The many-token arithmetic the full-size parity leg uses (tests/fullsize_parity.py): O1f (f32 blocked dots) equals O1
    (f64 dots) to f32 summation noise, and the C prompt step (one sequence, causal) equals the numpy prompt step --
    logits and the K/V it writes."""
    from oracle import cref
    cref.build()
    rng = np.random.default_rng(5)
    x = rng.normal(size=(37, 1024)).astype(np.float32)
    for t in (kq.GGML_Q4_K, kq.GGML_Q6_K):
        b = kq.quantize(rng.normal(0, 0.05, (45, 1024)).astype(np.float32), t)      # 45 rows: a ragged last row block
        o1 = kq.qmatmul_o1(x, b, t)
        assert np.abs(cref.qmatmul(x, b, t, o2=2) - o1).max() < 3e-6 * np.abs(o1).max()
    cfg = llama.LlamaConfig.tiny()
    W = llama.make_weights(cfg, seed=1234)
    M = llama.OracleLlama(cfg, W)
    seqs = [{"tokens": [int(t) for t in rng.integers(0, cfg.vocab, 45)], "block_table": [5, 2, 9]}]
    cache = M.new_cache(12)
    meta = O.prepare_prompt(seqs, cfg.block_size)
    ref = M.forward(meta, cache, is_prefill=True)[0]
    c_cache = [(np.zeros_like(k), np.zeros_like(v)) for k, v in cache]
    got = cref.CLlama(cfg, W).prefill(meta["input_ids"], meta["positions"], meta["slot_mapping"], c_cache)
    assert np.abs(got - ref).max() < 2e-3 * np.abs(ref).max()      # bf16 rounding flips of q/k/v between f32 and f64 sums
    assert int(got.argmax()) == int(ref.argmax())
    for (k1, v1), (k2, v2) in zip(cache, c_cache):
        assert np.abs(O.bf16_bits_to_f32(k1) - O.bf16_bits_to_f32(k2)).max() <= 2 ** -7 * np.abs(O.bf16_bits_to_f32(k1)).max()
        assert np.abs(O.bf16_bits_to_f32(v1) - O.bf16_bits_to_f32(v2)).max() <= 2 ** -7 * np.abs(O.bf16_bits_to_f32(v1)).max()
    # the accessor hands out the very bytes that were set
    cm = cref.CLlama(cfg, W)
    p, t, n, k = cm.qweight(1, 5)
    import ctypes
    blocks = np.ascontiguousarray(W["layers"][1]["w2"][1])
    assert (t, n, k) == (W["layers"][1]["w2"][0], blocks.shape[0], blocks.shape[1] * 256)
    assert ctypes.string_at(p, 64) == blocks.tobytes()[:64]


def test_c_oracle_rope_tables_follow_the_reference_at_long_positions():
    """calculate_default_inv_freq (rotary_emb.rs:14-19) takes the reciprocal in f32; a f64 reciprocal differs by one f32 ulp
    for 18 of Llama-3's 64 frequencies, which at position ~4096 is half a milliradian on the fastest pairs (found by the
    full-size parity leg).  The C twin must agree with the numpy restatement at such positions."""
    from oracle import cref
    cref.build()
    cfg = llama.LlamaConfig.tiny(max_seq=8192)
    cfg.rope_theta, cfg.head_dim, cfg.n_heads, cfg.n_kv_heads, cfg.hidden = 500000.0, 128, 2, 1, 256
    W = llama.make_weights(cfg, seed=77)
    M = llama.OracleLlama(cfg, W)
    rng = np.random.default_rng(3)
    L = 4099
    nblk = -(-L // cfg.block_size)
    seqs = [{"tokens": [int(t) for t in rng.integers(0, cfg.vocab, L)], "block_table": list(range(1, nblk + 1))}]
    cache = M.new_cache(nblk + 1)
    for kc, vc in cache:                                          # a random prefix instead of a 4k-token prompt step
        kc[...] = (rng.standard_normal(kc.shape).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
        vc[...] = (rng.standard_normal(vc.shape).astype(np.float32).view(np.uint32) >> 16).astype(np.uint16)
    meta = O.prepare_decode(seqs, cfg.block_size)
    c_cache = [(k.copy(), v.copy()) for k, v in cache]
    ref = M.forward(meta, cache)
    got = cref.CLlama(cfg, W).decode(meta, c_cache, o2=False)
    assert np.abs(got - ref).max() < 2e-4 * np.abs(ref).max()
    slot = int(meta["slot_mapping"][0])
    for (k1, _), (k2, _) in zip(cache, c_cache):                 # the new token's rotated K: identical up to a bf16 flip
        a = O.bf16_bits_to_f32(k1.reshape(-1, cfg.n_kv_heads * cfg.head_dim)[slot])
        b = O.bf16_bits_to_f32(k2.reshape(-1, cfg.n_kv_heads * cfg.head_dim)[slot])
        assert np.abs(a - b).max() <= 2 ** -7 * np.abs(a).max()
        assert (a != b).mean() < 0.05


def test_stablelm_oracle_plumbing_decode_equals_prefill():
    """BASELINE configs[0] (StableLM-3B bf16, CPU plumbing): the numpy restatement with LayerNorm + bias, head_dim 80,
    partial rotary (20 of 80 channels) and qkv bias runs a prompt step and greedy decode steps; a decode step equals
    the last row of the prompt step over the same tokens (cache write / gather / rotary positions are consistent)."""
    from oracle import dense_llama as DL
    from oracle import ops as O
    cfg = DL.DenseConfig.tiny_stablelm()
    full = DL.DenseConfig.stablelm_3b()
    assert (full.hidden, full.n_heads * full.head_dim, full.rotary_dim, full.intermediate) == (2560, 2560, 20, 6912)
    W = DL.make_weights(cfg)
    m = DL.OracleDenseLlama(cfg, W)
    rng = np.random.default_rng(0)
    seqs = [{"tokens": [int(t) for t in rng.integers(0, cfg.vocab, 19)], "block_table": [3, 7]},
            {"tokens": [int(t) for t in rng.integers(0, cfg.vocab, 5)], "block_table": [1]}]
    cache = m.new_cache(16)
    lg = m.forward(O.prepare_prompt(seqs, cfg.block_size), cache, is_prefill=True)
    assert lg.shape == (2, cfg.vocab) and np.isfinite(lg).all()
    for _ in range(2):
        for s, row in zip(seqs, lg):
            s["tokens"].append(int(row.argmax()))
        lg = m.forward(O.prepare_decode(seqs, cfg.block_size), cache)
    fresh = m.new_cache(16)
    again = m.forward(O.prepare_prompt(seqs, cfg.block_size), fresh, is_prefill=True)
    assert np.array_equal(again, lg)


@pytest.mark.parametrize("isa", ["avx2", "vnni"])
def test_c_twin_avx2_dot_products_equal_the_scalar_definition_bit_for_bit(isa):
    """oracle/oracle.c states the Q4_K x Q8_K and Q6_K x Q8_K integer dots three times (scalar definition; AVX2 and -- on hosts
    that have it -- AVX-512 VNNI `vpdpbusd` for the cpu_baseline's speed): random and extreme blocks (all codes 15 / 63 with
    activations +-127, the saturation corner of `maddubs`), identical float results"""
    import ctypes
    from oracle import cref
    cref.build()
    L = cref.lib()
    L.orc_isa.restype = ctypes.c_char_p
    L.orc_force_isa(0 if isa == "avx2" else -1)
    if isa == "vnni" and b"vnni" not in L.orc_isa():
        L.orc_force_isa(-1)
        pytest.skip("this host has no AVX-512 VNNI")
    vp, i32 = ctypes.c_void_p, ctypes.c_int32
    for name, args in (("orc_vec_dot_q4k_q8k", [vp, i32, vp, vp, vp]), ("orc_vec_dot_q4k_q8k_scalar", [vp, i32, vp, vp, vp]),
                       ("orc_vec_dot_q6k_q8k", [vp, i32, vp, vp]), ("orc_vec_dot_q6k_q8k_scalar", [vp, i32, vp, vp])):
        getattr(L, name).argtypes, getattr(L, name).restype = args, ctypes.c_float
    rng = np.random.default_rng(12)
    nb = 6
    for trial in range(60):
        xq = rng.integers(-127, 128, nb * 256).astype(np.int8)
        w4 = rng.integers(0, 256, nb * 144, dtype=np.uint8)
        w6 = rng.integers(0, 256, nb * 210, dtype=np.uint8)
        if trial % 5 == 0:                                          # extremes
            sign = 1 if trial % 10 == 0 else -1
            xq[:] = 127 * sign
            w4[:] = 0xFF
            w6[:] = 0xFF
            w6.reshape(nb, 210)[:, 192:208] = 0x7F if trial % 3 == 0 else 0x80      # scales +127 / -128
        for b in range(nb):                                         # finite f16 scales
            w4[b * 144:b * 144 + 4] = np.frombuffer(np.asarray([0.013, 0.007], np.float16).tobytes(), np.uint8)
            w6[b * 210 + 208:b * 210 + 210] = np.frombuffer(np.float16(0.011).tobytes(), np.uint8)
        xd = rng.uniform(0.01, 0.1, nb).astype(np.float32)
        xb = xq.astype(np.int32).reshape(-1, 16).sum(1).astype(np.int16)
        a = L.orc_vec_dot_q4k_q8k(w4.ctypes.data, nb, xd.ctypes.data, xq.ctypes.data, xb.ctypes.data)
        b_ = L.orc_vec_dot_q4k_q8k_scalar(w4.ctypes.data, nb, xd.ctypes.data, xq.ctypes.data, xb.ctypes.data)
        assert np.float32(a).tobytes() == np.float32(b_).tobytes(), (trial, a, b_)
        c = L.orc_vec_dot_q6k_q8k(w6.ctypes.data, nb, xd.ctypes.data, xq.ctypes.data)
        d = L.orc_vec_dot_q6k_q8k_scalar(w6.ctypes.data, nb, xd.ctypes.data, xq.ctypes.data)
        assert np.float32(c).tobytes() == np.float32(d).tobytes(), (trial, c, d)
    L.orc_force_isa(-1)
