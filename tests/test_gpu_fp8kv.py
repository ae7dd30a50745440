"""fp8 (OCP e4m3fn) KV cache (`--kvcache-dtype fp8`): cache write bit-exact vs the numpy conversion
(round-to-nearest-even, saturating), decode attention (MFMA partitions and the one-pass kernel) and prefill over the
quantised cache vs the oracle run on the dequantised cache."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from oracle import ops as O               # noqa: E402


@pytest.fixture(scope="module")
def cv(lib):
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X")
    import candle_vllm_amd.ops as ops
    return ops


def bf16_dev(vals_f32):
    bits = O.f32_to_bf16_bits(np.asarray(vals_f32, np.float32))
    return torch.from_numpy(np.ascontiguousarray(bits).view(np.int16)).cuda().view(torch.bfloat16)


def bf16_host(t):
    return O.bf16_bits_to_f32(t.detach().view(torch.int16).cpu().numpy().view(np.uint16))


def test_e4m3_cache_write_is_bit_exact(cv):
    rng = np.random.default_rng(0)
    NB, bs, Hkv, D, T = 6, 16, 2, 64, 41
    # values that hit ties, subnormals, saturation and negatives
    base = np.concatenate([rng.normal(0, 1, T * Hkv * D - 16), [0.0, -0.0, 1.0625, 1.1875, 447.9, 460.0, -1e6, 2 ** -9,
                                                               2 ** -10, 3 * 2 ** -10, 0.0175, -0.0175, 300.0, 17.0, 0.3, -0.3]])
    k = O.round_bf16(base.reshape(T, Hkv, D).astype(np.float32))
    v = O.round_bf16(rng.normal(0, 30, (T, Hkv, D)).astype(np.float32))
    ks, vs = O.kv_cache_shapes(NB, bs, Hkv, D, 1, False)
    kc = rng.integers(0, 256, ks).astype(np.uint8)
    vc = rng.integers(0, 256, vs).astype(np.uint8)
    slots = rng.permutation(NB * bs)[:T].astype(np.int64)
    slots[5] = -1
    kcd, vcd = torch.from_numpy(kc).cuda(), torch.from_numpy(vc).cuda()
    kd, vd, sd = bf16_dev(k), bf16_dev(v), torch.from_numpy(slots).cuda()       # keep the device buffers alive
    rc = cv.lib.mi355_reshape_and_cache_fp8(kd.data_ptr(), vd.data_ptr(), kcd.data_ptr(), vcd.data_ptr(),
                                            sd.data_ptr(), T, Hkv, D, bs, cv.KV_PAGED, 1.0, 1.0, 0)
    assert rc == 0
    torch.cuda.synchronize()
    O.reshape_and_cache_fp8(k, v, kc, vc, slots, False)
    assert np.array_equal(kcd.cpu().numpy(), kc)
    assert np.array_equal(vcd.cpu().numpy(), vc)


@pytest.mark.parametrize("H,Hkv,D,bs,ctxs,ps", [(8, 2, 128, 64, [300, 64, 129], None), (4, 4, 64, 16, [40, 7], 0),
                                                (8, 2, 128, 16, [200, 33], 32), (4, 2, 80, 16, [50], 0),
                                                (8, 2, 128, 64, [700, 300, 64, 257], 256), (8, 2, 128, 16, [1100, 33], 512)])
def test_decode_attention_over_fp8_cache(cv, H, Hkv, D, bs, ctxs, ps):
    _decode_case(cv, H, Hkv, D, bs, ctxs, ps)


@pytest.mark.parametrize("H,Hkv,bs,ctxs,force", [(8, 2, 64, [300, 64, 129, 700], True), (28, 4, 16, None, False), (16, 1, 32, [513, 64, 1, 90, 2000], True),
                                                 (32, 8, 32, None, False), (6, 2, 16, [65, 127, 128, 129], True)])
def test_decode_attention_over_fp8_cache_balanced_stream(cv, H, Hkv, bs, ctxs, force):
    """round 5: >= 64 (sequence, kv head) pairs at partition size 64 stream the e4m3fn cache through the LDS-DMA ring in 16 KiB stages
    (`paged_attn_stream_kernel<3, true, true>`): ragged contexts whose last stage is masked, 1..16 query heads per kv head, block sizes
    16 / 32 / 64, stale NaN bytes in unused slots; small launches are forced onto it (tuning key 44 = 3).  Against the oracle over the
    dequantised cache, and against the 32-token MFMA partitions (key 44 = 0) to the rounding of the bf16 output."""
    from candle_vllm_amd import tuning
    if ctxs is None:                                               # natural: 16 or 8 sequences x 4 or 8 kv heads = 64 pairs
        rng = np.random.default_rng(H)
        ctxs = [int(c) for c in rng.integers(1, 1500, 64 // Hkv)]
    with tuning(44, 3 if force else 1):
        got = _decode_case(cv, H, Hkv, 128, bs, ctxs, 64, seed=7)
    with tuning(44, 0):
        base = _decode_case(cv, H, Hkv, 128, bs, ctxs, 32, seed=7)
    assert np.abs(got - base).max() <= 2.0 ** -7 * max(1.0, np.abs(base).max())


def _decode_case(cv, H, Hkv, D, bs, ctxs, ps, seed=0):
    if D % 16:
        pytest.skip("x = 16 layout needs head_dim % 16 == 0")
    rng = np.random.default_rng(D + bs + len(ctxs) + seed)
    B = len(ctxs)
    nblk = [-(-c // bs) for c in ctxs]
    NB = sum(nblk) + 2
    ids = rng.permutation(NB)[: sum(nblk)]
    ks, vs = O.kv_cache_shapes(NB, bs, Hkv, D, 1, False)
    kc = np.zeros(ks, np.uint8)
    vc = np.zeros(vs, np.uint8)
    # unused slots hold arbitrary bytes, including the NaN encodings 0x7F / 0xFF
    kc[:] = rng.integers(0, 256, ks)
    vc[:] = rng.integers(0, 256, vs)
    seqs, o = [], 0
    pa = cv.PagedAttention(H, D, 1.0 / np.sqrt(D), Hkv, fp8_kvcache=True)
    kcd, vcd = torch.from_numpy(kc).cuda(), torch.from_numpy(vc).cuda()
    for b, c in enumerate(ctxs):
        table = ids[o:o + nblk[b]].tolist()
        o += nblk[b]
        seqs.append({"tokens": list(range(c)), "block_table": table})
        kk = O.round_bf16(rng.normal(0, 1.5, (c - 1, Hkv, D)).astype(np.float32))
        vv = O.round_bf16(rng.normal(0, 1.5, (c - 1, Hkv, D)).astype(np.float32))
        slots = np.array([table[j // bs] * bs + j % bs for j in range(c - 1)], np.int64)
        O.reshape_and_cache_fp8(kk, vv, kc, vc, slots, False)
    kcd.copy_(torch.from_numpy(kc)); vcd.copy_(torch.from_numpy(vc))
    meta = O.prepare_decode(seqs, bs)
    im = cv.InputMetadata.from_oracle_meta(meta, "cuda")
    q = O.round_bf16(rng.normal(0, 1, (B, H, D)).astype(np.float32))
    k = O.round_bf16(rng.normal(0, 1.5, (B, Hkv, D)).astype(np.float32))
    v = O.round_bf16(rng.normal(0, 1.5, (B, Hkv, D)).astype(np.float32))
    out = pa.forward(bf16_dev(q), bf16_dev(k), bf16_dev(v), None, kcd, vcd, im, partition_size=ps)
    O.reshape_and_cache_fp8(k, v, kc, vc, meta["slot_mapping"], False)
    assert np.array_equal(kcd.cpu().numpy(), kc) and np.array_equal(vcd.cpu().numpy(), vc)
    kb, vb = O.fp8_cache_as_bf16_bits(kc, vc, False)
    ref = O.paged_attention_decode(q, kb, vb, meta["block_tables"], meta["context_lens"], 1.0 / np.sqrt(D), False)
    got = bf16_host(out)
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() <= 2e-2 * max(1.0, np.abs(ref).max())     # bf16 P and output rounding
    return got


@pytest.mark.parametrize("generic", [0, 1])
@pytest.mark.parametrize("H,Hkv,D,bs,lens,cached", [(4, 2, 64, 16, [37, 20], [0, 24]), (8, 2, 128, 64, [150, 33, 5], [0, 70, 129]),
                                                    # head size 128 (round 6): the blocks as bf16 in a temporary cache + the LDS-fed kernel; 16- and 32-token
                                                    # blocks put several blocks -- some past the context, their table entries unused -- into one 64-token stage
                                                    (8, 2, 128, 16, [150, 33, 5, 200], [0, 70, 129, 1000]), (4, 1, 128, 32, [70, 1, 300], [500, 31, 0])])
def test_prefill_over_fp8_cache(cv, H, Hkv, D, bs, lens, cached, generic):
    """prefill over the e4m3 cache (attention.rs:574,896): the MFMA flash kernel (bytes -> bf16 fragments on the way in; several
    128-query tiles, cached prefixes that cross blocks) and the generic kernel (tuning key 43) against the oracle on the
    e4m3-rounded K / V, same bound"""
    from candle_vllm_amd import tuning
    rng = np.random.default_rng(9)
    pa = cv.PagedAttention(H, D, 1.0 / np.sqrt(D), Hkv, fp8_kvcache=True)
    ctx = [c + l for c, l in zip(cached, lens)]
    nblk = [-(-c // bs) for c in ctx]
    NB = sum(nblk) + 1
    ids = rng.permutation(NB)[: sum(nblk)]
    ks, vs = O.kv_cache_shapes(NB, bs, Hkv, D, 1, False)
    kc = rng.integers(0, 256, ks).astype(np.uint8)
    vc = rng.integers(0, 256, vs).astype(np.uint8)
    seqs, o, k_all, v_all = [], 0, [], []
    for b, c in enumerate(ctx):
        table = ids[o:o + nblk[b]].tolist()
        o += nblk[b]
        seqs.append({"tokens": list(range(c)), "block_table": table})
        k_all.append(O.round_bf16(rng.normal(0, 1.5, (c, Hkv, D)).astype(np.float32)))
        v_all.append(O.round_bf16(rng.normal(0, 1.5, (c, Hkv, D)).astype(np.float32)))
        if cached[b]:
            slots = np.array([table[j // bs] * bs + j % bs for j in range(cached[b])], np.int64)
            O.reshape_and_cache_fp8(k_all[b][:cached[b]], v_all[b][:cached[b]], kc, vc, slots, False)
    meta = O.prepare_prompt(seqs, bs, cached)
    im = cv.InputMetadata.from_oracle_meta(meta, "cuda", is_prefill=True)
    q = [O.round_bf16(rng.normal(0, 1, (l, H, D)).astype(np.float32)) for l in lens]
    kcd, vcd = torch.from_numpy(kc).cuda(), torch.from_numpy(vc).cuda()
    kn = np.concatenate([k_all[b][cached[b]:] for b in range(len(lens))])
    vn = np.concatenate([v_all[b][cached[b]:] for b in range(len(lens))])
    with tuning(47, 1 | (2 if generic else 0)):                  # key 47 bit 1: fp8-cache prompt attention on the generic kernel
        out = bf16_host(pa.forward(bf16_dev(np.concatenate(q)), bf16_dev(kn), bf16_dev(vn), None, kcd, vcd, im))
    O.reshape_and_cache_fp8(kn, vn, kc, vc, meta["slot_mapping"], False)
    assert np.array_equal(kcd.cpu().numpy(), kc)
    o = 0
    for b, l in enumerate(lens):
        kq = O.e4m3fn_to_f32(O.f32_to_e4m3fn(k_all[b]))
        vq = O.e4m3fn_to_f32(O.f32_to_e4m3fn(v_all[b]))
        ref = O.prefill_attention(q[b], kq, vq, 1.0 / np.sqrt(D), cached=cached[b])
        assert np.abs(out[o:o + l] - ref).max() <= 2e-2 * max(1.0, np.abs(ref).max())
        o += l
