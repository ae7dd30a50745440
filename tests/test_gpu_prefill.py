"""GPU parity tests for K4 (prefill attention) through the C ABI vs the numpy oracle (softmax in f64).
Tolerance: flash-attention arithmetic (P rounded to 16 bit, f32 accumulation) vs exact softmax -> 2e-2 of the
output scale for bf16, 4e-3 for f16, stated next to the assert; the output itself is a 16-bit value."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from oracle import ops as O                # noqa: E402
from oracle import gptq as G               # noqa: E402  (16-bit helpers)

TD = {"bf16": torch.bfloat16, "f16": torch.float16}
TOL = {"bf16": 2e-2, "f16": 4e-3}


@pytest.fixture(scope="module")
def cv(lib):
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X (torch.cuda.is_available() is False)")
    import candle_vllm_amd.ops as ops
    return ops


def dev16(a, dt):
    return torch.from_numpy(np.ascontiguousarray(G.to_bits(a, dt)).view(np.int16)).cuda().view(TD[dt])


def host16(t, dt):
    return G.from_bits(t.detach().view(torch.int16).cpu().numpy().view(np.uint16), dt)


def make_case(rng, lens, cached, H, Hkv, D, bs, dt, flash):
    """sequences with `cached[i]` tokens already in the cache and lens[i] new tokens; shuffled block ids"""
    n = len(lens)
    ctx = [c + l for c, l in zip(cached, lens)]
    nblk = [-(-c // bs) for c in ctx]
    NB = sum(nblk) + 3
    ids = rng.permutation(NB)[: sum(nblk)]
    tables, o = [], 0
    for b in nblk:
        tables.append(ids[o:o + b].tolist())
        o += b
    ks, vs = O.kv_cache_shapes(NB, bs, Hkv, D, 2, flash)
    kc = rng.integers(0, 65536, ks).astype(np.uint16)          # arbitrary bits (incl. NaN patterns) in unused slots
    vc = rng.integers(0, 65536, vs).astype(np.uint16)
    k_all = [G.round_dt(rng.normal(0, 1, (c, Hkv, D)), dt) for c in ctx]
    v_all = [G.round_dt(rng.normal(0, 1, (c, Hkv, D)), dt) for c in ctx]
    q = [G.round_dt(rng.normal(0, 1, (l, H, D)), dt) for l in lens]
    # the whole context (prefix + chunk) is in the cache, as after reshape_and_cache
    for i in range(n):
        slots = np.array([tables[i][j // bs] * bs + j % bs for j in range(ctx[i])], np.int64)
        O.reshape_and_cache(G.to_bits(k_all[i], dt), G.to_bits(v_all[i], dt), kc, vc, slots, flash)
    seqs = [{"tokens": list(range(ctx[i])), "block_table": tables[i]} for i in range(n)]
    meta = O.prepare_prompt(seqs, bs, cached)
    return q, k_all, v_all, kc, vc, meta


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("H,Hkv,D", [(8, 2, 128), (4, 4, 64), (4, 2, 80)])
@pytest.mark.parametrize("lens", [[1], [37], [130, 5, 64], [257]])
def test_prefill_no_cache(cv, dt, H, Hkv, D, lens):
    rng = np.random.default_rng(sum(lens) + D)
    q, k_all, v_all, kc, vc, meta = make_case(rng, lens, [0] * len(lens), H, Hkv, D, 16, dt, True)
    scale = 1.0 / np.sqrt(D)
    pa = cv.PagedAttention(H, D, scale, Hkv)
    im = cv.InputMetadata.from_oracle_meta(meta, "cuda", is_prefill=True)
    kcat, vcat = np.concatenate(k_all), np.concatenate(v_all)
    out = pa.prefill(dev16(np.concatenate(q), dt), dev16(kcat, dt), dev16(vcat, dt), None, None, im)
    got = host16(out, dt)
    o = 0
    for i, l in enumerate(lens):
        ref = O.prefill_attention(q[i], k_all[i], v_all[i], scale, rnd=lambda a: G.round_dt(a, dt))
        assert np.abs(got[o:o + l] - ref).max() <= TOL[dt] * max(1.0, np.abs(ref).max()), f"seq {i}"
        o += l


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("flash", [True, False])
@pytest.mark.parametrize("H,Hkv,D,bs", [(8, 2, 128, 64), (4, 4, 64, 16), (4, 2, 80, 16)])
def test_prefill_cached_prefix(cv, dt, flash, H, Hkv, D, bs):
    rng = np.random.default_rng(D + bs + int(flash))
    lens, cached = [70, 3, 129], [50, 0, 200]                     # chunked prefill / prefix-cache hit / fresh
    q, k_all, v_all, kc, vc, meta = make_case(rng, lens, cached, H, Hkv, D, bs, dt, flash)
    scale = 1.0 / np.sqrt(D)
    pa = cv.PagedAttention(H, D, scale, Hkv)
    im = cv.InputMetadata.from_oracle_meta(meta, "cuda", is_prefill=True)
    assert im.max_seqlen_k > im.max_seqlen_q
    kcd = torch.from_numpy(kc.view(np.int16)).cuda().view(TD[dt])
    vcd = torch.from_numpy(vc.view(np.int16)).cuda().view(TD[dt])
    out = pa.prefill(dev16(np.concatenate(q), dt), None, None, kcd, vcd, im)
    got = host16(out, dt)
    assert np.isfinite(got).all()
    o = 0
    for i, l in enumerate(lens):
        ref = O.prefill_attention(q[i], k_all[i], v_all[i], scale, cached=cached[i], rnd=lambda a: G.round_dt(a, dt))
        assert np.abs(got[o:o + l] - ref).max() <= TOL[dt] * max(1.0, np.abs(ref).max()), f"seq {i}"
        o += l


def test_prefill_softcap_and_decode_consistency(cv):
    """softcap path; and the last prefill row equals the decode kernel's answer for the same context"""
    dt, H, Hkv, D, bs = "bf16", 8, 2, 128, 16
    rng = np.random.default_rng(5)
    lens = [90]
    q, k_all, v_all, kc, vc, meta = make_case(rng, lens, [0], H, Hkv, D, bs, dt, False)
    scale = 1.0 / np.sqrt(D)
    pa = cv.PagedAttention(H, D, scale, Hkv)
    im = cv.InputMetadata.from_oracle_meta(meta, "cuda", is_prefill=True)
    out = pa.prefill(dev16(q[0], dt), dev16(k_all[0], dt), dev16(v_all[0], dt), None, None, im, softcapping=30.0)
    ref = O.prefill_attention(q[0], k_all[0], v_all[0], scale, softcap=30.0)
    assert np.abs(host16(out, dt) - ref).max() <= TOL[dt] * max(1.0, np.abs(ref).max())
    out2 = host16(pa.prefill(dev16(q[0], dt), dev16(k_all[0], dt), dev16(v_all[0], dt), None, None, im), dt)
    kcd = torch.from_numpy(kc.view(np.int16)).cuda().view(TD[dt])
    vcd = torch.from_numpy(vc.view(np.int16)).cuda().view(TD[dt])
    dmeta = O.prepare_decode([{"tokens": list(range(90)), "block_table": meta["block_tables"][0].tolist()}], bs)
    dm = cv.InputMetadata.from_oracle_meta(dmeta, "cuda")
    dec = host16(pa.decode(dev16(q[0][-1:], dt), kcd, vcd, dm), dt)
    assert np.abs(dec[0] - out2[-1]).max() <= TOL[dt]


def test_prefill_mixed_batch_cached_and_fresh_with_equal_maxima(cv):
    """a batch where max_seqlen_k == max_seqlen_q although ONE sequence has a cached prefix (16 cached + 16
    new beside 64 fresh tokens).  The reference decides `use_cached_kv` per sequence (inputs.rs:132-143); the wrapper
    must take the cache path, otherwise the cached prefix is silently dropped from the first sequence's attention."""
    dt, H, Hkv, D, bs = "bf16", 8, 2, 128, 16
    rng = np.random.default_rng(77)
    lens, cached = [16, 64], [16, 0]
    q, k_all, v_all, kc, vc, meta = make_case(rng, lens, cached, H, Hkv, D, bs, dt, False)
    scale = 1.0 / np.sqrt(D)
    pa = cv.PagedAttention(H, D, scale, Hkv)
    im = cv.InputMetadata.from_oracle_meta(meta, "cuda", is_prefill=True)
    assert im.max_seqlen_k == im.max_seqlen_q                      # the batch-level test cannot see the prefix
    kcd = torch.from_numpy(kc.view(np.int16)).cuda().view(TD[dt])
    vcd = torch.from_numpy(vc.view(np.int16)).cuda().view(TD[dt])
    # the chunk's own k / v are handed over too (as the model does): the wrapper must still read the cache
    knew = np.concatenate([k_all[i][cached[i]:] for i in range(2)])
    vnew = np.concatenate([v_all[i][cached[i]:] for i in range(2)])
    out = pa.prefill(dev16(np.concatenate(q), dt), dev16(knew, dt), dev16(vnew, dt), kcd, vcd, im)
    got = host16(out, dt)
    o = 0
    for i, l in enumerate(lens):
        ref = O.prefill_attention(q[i], k_all[i], v_all[i], scale, cached=cached[i], rnd=lambda a: G.round_dt(a, dt))
        assert np.abs(got[o:o + l] - ref).max() <= TOL[dt] * max(1.0, np.abs(ref).max()), f"seq {i}"
        o += l


@pytest.mark.parametrize("H,Hkv,bs", [(8, 2, 64), (32, 8, 16), (28, 4, 32), (4, 4, 64), (8, 1, 16)])
def test_prefill_lds_dma_kernel(cv, H, Hkv, bs):
    """the default since round 4: prompt attention with K / V through the LDS ring (prefill_attn_lds_kernel): 64 queries x the heads of a GQA
    group per workgroup -- chunked prefill, prefix hit, fresh prompts, blocks that end inside a stage, GQA groups of 1 / 4 / 7 / 8
    heads; against the oracle at the product kernel's bound AND against the product kernel itself (same hi + lo probabilities:
    the two agree to accumulation noise)"""
    from candle_vllm_amd import tuning
    dt, D = "bf16", 128
    rng = np.random.default_rng(H + bs)
    lens, cached = [70, 3, 129, 64, 200, 1], [50, 0, 200, 64, 0, 300]
    q, k_all, v_all, kc, vc, meta = make_case(rng, lens, cached, H, Hkv, D, bs, dt, False)
    scale = 1.0 / np.sqrt(D)
    pa = cv.PagedAttention(H, D, scale, Hkv)
    im = cv.InputMetadata.from_oracle_meta(meta, "cuda", is_prefill=True)
    kcd = torch.from_numpy(kc.view(np.int16)).cuda().view(TD[dt])
    vcd = torch.from_numpy(vc.view(np.int16)).cuda().view(TD[dt])
    qd = dev16(np.concatenate(q), dt)
    with tuning(47, 0):                                               # the register-fed kernel (A/B switch)
        base = host16(pa.prefill(qd, None, None, kcd, vcd, im), dt)
        soft_base = host16(pa.prefill(qd, None, None, kcd, vcd, im, 30.0), dt)
    got = host16(pa.prefill(qd, None, None, kcd, vcd, im), dt)
    soft = host16(pa.prefill(qd, None, None, kcd, vcd, im, 30.0), dt)
    assert np.isfinite(got).all() and np.isfinite(soft).all()
    o = 0
    for i, l in enumerate(lens):
        ref = O.prefill_attention(q[i], k_all[i], v_all[i], scale, cached=cached[i], rnd=lambda a: G.round_dt(a, dt))
        assert np.abs(got[o:o + l] - ref).max() <= TOL[dt] * max(1.0, np.abs(ref).max()), f"seq {i}"
        o += l
    assert np.abs(got - base).max() <= 2 ** -7 * max(1.0, np.abs(base).max())        # one bf16 ulp of the largest output
    assert np.abs(soft - soft_base).max() <= 2 ** -7 * max(1.0, np.abs(soft_base).max())



@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("flash", [True, False])
@pytest.mark.parametrize("H,Hkv,D,bs,window", [(8, 2, 128, 64, 100), (4, 4, 64, 16, 7), (4, 2, 80, 16, 33), (8, 2, 128, 16, 4096)])
def test_sliding_window_prompt_and_decode(cv, dt, flash, H, Hkv, D, bs, window):
    """`sliding_window` of PagedAttention::new (attention.rs:566-575,888-897): prompt steps with cached prefixes, without a cache, and the
    decode step, against the oracle's windowed attention (anchored to transformers' sliding-window mask in test_cpu_third_party.py).
    The generic kernels keep f32 softmax: same bound as the other prompt tests.  A window larger than every context = no window."""
    rng = np.random.default_rng(D + bs + window + int(flash))
    lens, cached = [70, 3, 129], [50, 0, 200]
    q, k_all, v_all, kc, vc, meta = make_case(rng, lens, cached, H, Hkv, D, bs, dt, flash)
    scale = 1.0 / np.sqrt(D)
    pa = cv.PagedAttention(H, D, scale, Hkv, sliding_window=window)
    im = cv.InputMetadata.from_oracle_meta(meta, "cuda", is_prefill=True)
    kcd = torch.from_numpy(kc.view(np.int16)).cuda().view(TD[dt])
    vcd = torch.from_numpy(vc.view(np.int16)).cuda().view(TD[dt])
    got = host16(pa.prefill(dev16(np.concatenate(q), dt), None, None, kcd, vcd, im), dt)
    assert np.isfinite(got).all()
    o = 0
    for i, l in enumerate(lens):
        ref = O.prefill_attention(q[i], k_all[i], v_all[i], scale, cached=cached[i], rnd=lambda a: G.round_dt(a, dt), sliding_window=window)
        assert np.abs(got[o:o + l] - ref).max() <= TOL[dt] * max(1.0, np.abs(ref).max()), f"seq {i}"
        if window < cached[i] + l:                                 # the window really cuts keys off: differs from plain causal attention
            full = O.prefill_attention(q[i], k_all[i], v_all[i], scale, cached=cached[i], rnd=lambda a: G.round_dt(a, dt))
            assert np.abs(full - ref).max() > 1e-3
        o += l
    # no cache: the chunk's own keys
    q2, k2, v2, _, _, meta2 = make_case(rng, [90, 5], [0, 0], H, Hkv, D, bs, dt, True)
    im2 = cv.InputMetadata.from_oracle_meta(meta2, "cuda", is_prefill=True)
    got2 = host16(pa.prefill(dev16(np.concatenate(q2), dt), dev16(np.concatenate(k2), dt), dev16(np.concatenate(v2), dt), None, None, im2), dt)
    o = 0
    for i, l in enumerate([90, 5]):
        ref = O.prefill_attention(q2[i], k2[i], v2[i], scale, rnd=lambda a: G.round_dt(a, dt), sliding_window=window)
        assert np.abs(got2[o:o + l] - ref).max() <= TOL[dt] * max(1.0, np.abs(ref).max()), f"fresh seq {i}"
        o += l
    # decode: one query per sequence over its whole cached context
    ctx = [c + l for c, l in zip(cached, lens)]
    seqs = [{"tokens": list(range(ctx[i])), "block_table": meta["block_tables"][i][: -(-ctx[i] // bs)].tolist()} for i in range(len(lens))]
    dm = cv.InputMetadata.from_oracle_meta(O.prepare_decode(seqs, bs), "cuda")
    qd = np.stack([q[i][-1] for i in range(len(lens))])
    dec = host16(pa.decode(dev16(qd, dt), kcd, vcd, dm), dt)
    for i in range(len(lens)):
        ref = O.prefill_attention(q[i][-1:], k_all[i], v_all[i], scale, cached=ctx[i] - 1, rnd=lambda a: G.round_dt(a, dt), sliding_window=window)[0]
        assert np.abs(dec[i] - ref).max() <= TOL[dt] * max(1.0, np.abs(ref).max()), f"decode seq {i}"
