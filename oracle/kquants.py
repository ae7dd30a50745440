"""CPU oracle (TEST INFRASTRUCTURE ONLY) -- GGUF k-quant block formats in numpy.

PARITY UNPINNED: the arithmetic restated here lives in the reference's un-vendored
candle fork (guoqingbao/candle v0.8.3 rev cafd231, `candle-core/src/quantized/k_quants.rs`,
Cargo.toml:24-25 of the reference) which is absent from /root/reference, and the
reference holds no golden vectors for it.  The formats follow the
published ggml k-quant layout; the reference call sites that fix
the semantics are `QMatMul::forward` at src/openai/models/layers/attention.rs:920-922,1004,
src/openai/models/quantized_llama.rs:33-37 and src/openai/models/linear.rs:765-806.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Two oracles are provided for the quantized mat-vec:
  O1  dequantize -> f64 dot  (what an exact dequant kernel must equal; our parity target)
  O2  candle-CPU-faithful: activations quantised to Q8_K per 256, integer dot
      (`vec_dot_q4k_q8k` style); differs from O1 by ~0.5-1 % and is reported, not gated.
"""
import numpy as np

QK_K = 256
Q4_K_BLOCK_BYTES = 144   # f16 d; f16 dmin; u8 scales[12]; u8 qs[128]
Q6_K_BLOCK_BYTES = 210   # u8 ql[128]; u8 qh[64]; i8 scales[16]; f16 d
Q8_0_BLOCK_BYTES = 34    # f16 d; i8 qs[32]


# --------------------------------------------------------------------------- Q4_K
def _unpack_scales_q4k(scales):
    """scales: uint8 [..., 12] -> (sc, m) uint8 [..., 8]   (get_scale_min_k4)."""
    s = scales.astype(np.uint8)
    sc = np.empty(s.shape[:-1] + (8,), np.uint8)
    m = np.empty_like(sc)
    for j in range(4):
        sc[..., j] = s[..., j] & 63
        m[..., j] = s[..., j + 4] & 63
    for j in range(4, 8):
        sc[..., j] = (s[..., j + 4] & 0x0F) | ((s[..., j - 4] >> 6) << 4)
        m[..., j] = (s[..., j + 4] >> 4) | ((s[..., j] >> 6) << 4)
    return sc, m


def _pack_scales_q4k(sc, m):
    """inverse of _unpack_scales_q4k; sc, m uint8 [..., 8] (<64) -> uint8 [..., 12]."""
    sc = sc.astype(np.uint8)
    m = m.astype(np.uint8)
    out = np.zeros(sc.shape[:-1] + (12,), np.uint8)
    for j in range(4):
        out[..., j] = sc[..., j] | ((sc[..., j + 4] >> 4) << 6)
        out[..., j + 4] = m[..., j] | ((m[..., j + 4] >> 4) << 6)
        out[..., j + 8] = (sc[..., j + 4] & 0x0F) | ((m[..., j + 4] & 0x0F) << 4)
    return out


def dequantize_q4_k(blocks):
    """blocks: uint8 [..., nblk, 144] -> float32 [..., nblk*256].

    w[64g + l]      = d*sc[2g]  *(qs[32g+l] & 0xF) - dmin*m[2g]
    w[64g + 32 + l] = d*sc[2g+1]*(qs[32g+l] >> 4)  - dmin*m[2g+1]
    Format [EXT ggml k-quants, SURVEY.md App. C]; semantics fixed by the QMatMul::forward call sites src/openai/models/layers/attention.rs:920-922 and src/openai/models/quantized_llama.rs:33-37 (candle k_quants.rs is un-vendored)."""
    b = np.ascontiguousarray(blocks, dtype=np.uint8)
    assert b.shape[-1] == Q4_K_BLOCK_BYTES
    lead = b.shape[:-1]
    d = b[..., 0:2].copy().view(np.float16)[..., 0].astype(np.float32)
    dmin = b[..., 2:4].copy().view(np.float16)[..., 0].astype(np.float32)
    sc, m = _unpack_scales_q4k(b[..., 4:16])
    qs = b[..., 16:144].reshape(lead + (4, 32))
    lo = (qs & 0x0F).astype(np.float32)
    hi = (qs >> 4).astype(np.float32)
    q = np.stack([lo, hi], axis=-2).reshape(lead + (8, 32))       # sub-block j = 2g + {0,1}
    dsc = (d[..., None] * sc.astype(np.float32))[..., None]
    dm = (dmin[..., None] * m.astype(np.float32))[..., None]
    w = dsc * q - dm
    return w.reshape(lead[:-1] + (lead[-1] * QK_K,)).astype(np.float32)


def quantize_q4_k(w):
    """w: float [..., K] (K % 256 == 0) -> uint8 [..., K/256, 144].

    A plain min/max quantiser producing VALID Q4_K blocks.  It is NOT ggml's
    search-based `quantize_row_q4_K` (make_qkx2_quants); the reference never quantises on the
    decode hot path (only ISQ / TP re-shard, out of scope), so only the *format* matters.
    Format [EXT ggml k-quants, SURVEY.md App. C]; semantics fixed by the QMatMul::forward call sites src/openai/models/layers/attention.rs:920-922 and src/openai/models/quantized_llama.rs:33-37 (candle k_quants.rs is un-vendored)."""
    w = np.asarray(w, dtype=np.float32)
    K = w.shape[-1]
    assert K % QK_K == 0
    lead = w.shape[:-1]
    x = w.reshape(lead + (K // QK_K, 8, 32))
    mn = np.minimum(x.min(-1), 0.0)                 # ggml clamps the min to <= 0
    mx = x.max(-1)
    scale = (mx - mn) / 15.0
    mins = -mn
    max_scale = scale.max(-1)
    max_min = mins.max(-1)
    d = (max_scale / 63.0).astype(np.float16)
    dmin = (max_min / 63.0).astype(np.float16)
    df = d.astype(np.float32)[..., None]
    dmf = dmin.astype(np.float32)[..., None]
    with np.errstate(divide="ignore", invalid="ignore"):
        sc = np.where(df > 0, np.rint(scale / df), 0).clip(0, 63).astype(np.uint8)
        m = np.where(dmf > 0, np.rint(mins / dmf), 0).clip(0, 63).astype(np.uint8)
        dsc = (df * sc.astype(np.float32))[..., None]
        dm = (dmf * m.astype(np.float32))[..., None]
        q = np.where(dsc > 0, np.rint((x + dm) / dsc), 0).clip(0, 15).astype(np.uint8)
    q = q.reshape(lead + (K // QK_K, 4, 2, 32))
    qs = (q[..., 0, :] | (q[..., 1, :] << 4)).reshape(lead + (K // QK_K, 128))
    out = np.zeros(lead + (K // QK_K, Q4_K_BLOCK_BYTES), np.uint8)
    out[..., 0:2] = d[..., None].view(np.uint8)
    out[..., 2:4] = dmin[..., None].view(np.uint8)
    out[..., 4:16] = _pack_scales_q4k(sc, m)
    out[..., 16:144] = qs
    return out


# --------------------------------------------------------------------------- Q6_K
def dequantize_q6_k(blocks):
    """blocks: uint8 [..., nblk, 210] -> float32 [..., nblk*256]  (SURVEY.md App. C)."""
    b = np.ascontiguousarray(blocks, dtype=np.uint8)
    assert b.shape[-1] == Q6_K_BLOCK_BYTES
    lead = b.shape[:-1]
    ql = b[..., 0:128].reshape(lead + (2, 64))
    qh = b[..., 128:192].reshape(lead + (2, 32))
    sc = b[..., 192:208].copy().view(np.int8).reshape(lead + (2, 8)).astype(np.float32)
    d = b[..., 208:210].copy().view(np.float16)[..., 0].astype(np.float32)
    q1 = (ql[..., 0:32] & 0x0F) | (((qh >> 0) & 3) << 4)
    q2 = (ql[..., 32:64] & 0x0F) | (((qh >> 2) & 3) << 4)
    q3 = (ql[..., 0:32] >> 4) | (((qh >> 4) & 3) << 4)
    q4 = (ql[..., 32:64] >> 4) | (((qh >> 6) & 3) << 4)
    q = np.stack([q1, q2, q3, q4], axis=-2).astype(np.int32) - 32     # [..., 2, 4, 32]
    # element (n, t, l): index 128n + 32t + l ; scale index 8n + 2t + l//16
    q = q.reshape(lead + (2, 4, 2, 16)).astype(np.float32)
    scv = sc.reshape(lead + (2, 4, 2))[..., None]
    w = d[..., None, None, None, None] * scv * q
    return w.reshape(lead[:-1] + (lead[-1] * QK_K,)).astype(np.float32)


def quantize_q6_k(w):
    """w: float [..., K] -> uint8 [..., K/256, 210]; plain absmax quantiser (valid format).
    Format [EXT ggml k-quants, SURVEY.md App. C]; semantics fixed by the QMatMul::forward call sites src/openai/models/layers/attention.rs:920-922 and src/openai/models/quantized_llama.rs:33-37 (candle k_quants.rs is un-vendored)."""
    w = np.asarray(w, dtype=np.float32)
    K = w.shape[-1]
    assert K % QK_K == 0
    lead = w.shape[:-1]
    nb = K // QK_K
    x = w.reshape(lead + (nb, 16, 16))
    amax_idx = np.abs(x).argmax(-1)
    amax = np.take_along_axis(x, amax_idx[..., None], -1)[..., 0]
    scale = amax / -32.0                                   # ggml: iscale = -32/max
    smax_idx = np.abs(scale).argmax(-1)
    smax = np.take_along_axis(scale, smax_idx[..., None], -1)[..., 0]
    d = (smax / -128.0).astype(np.float16)                 # so that the extreme scale -> -128
    df = d.astype(np.float32)[..., None]
    with np.errstate(divide="ignore", invalid="ignore"):
        sc = np.where(df != 0, np.rint(scale / df), 0).clip(-128, 127).astype(np.int8)
        dsc = (df * sc.astype(np.float32))[..., None]
        q = np.where(dsc != 0, np.rint(x / dsc), 0).clip(-32, 31).astype(np.int32) + 32
    q = q.astype(np.uint8).reshape(lead + (nb, 2, 4, 32))  # (n, t, l)
    lo = q & 0x0F
    hi = q >> 4
    ql = np.empty(lead + (nb, 2, 64), np.uint8)
    ql[..., 0:32] = lo[..., 0, :] | (lo[..., 2, :] << 4)
    ql[..., 32:64] = lo[..., 1, :] | (lo[..., 3, :] << 4)
    qh = hi[..., 0, :] | (hi[..., 1, :] << 2) | (hi[..., 2, :] << 4) | (hi[..., 3, :] << 6)
    out = np.zeros(lead + (nb, Q6_K_BLOCK_BYTES), np.uint8)
    out[..., 0:128] = ql.reshape(lead + (nb, 128))
    out[..., 128:192] = qh.reshape(lead + (nb, 64))
    out[..., 192:208] = sc.reshape(lead + (nb, 16)).view(np.uint8)
    out[..., 208:210] = d[..., None].view(np.uint8)
    return out


# --------------------------------------------------------------------------- Q8_0
def dequantize_q8_0(blocks):
    """Format [EXT ggml k-quants, SURVEY.md App. C]; semantics fixed by the QMatMul::forward call sites src/openai/models/layers/attention.rs:920-922 and src/openai/models/quantized_llama.rs:33-37 (candle k_quants.rs is un-vendored)."""
    b = np.ascontiguousarray(blocks, dtype=np.uint8)
    assert b.shape[-1] == Q8_0_BLOCK_BYTES
    lead = b.shape[:-1]
    d = b[..., 0:2].copy().view(np.float16)[..., 0].astype(np.float32)
    q = b[..., 2:34].copy().view(np.int8).astype(np.float32)
    return (d[..., None] * q).reshape(lead[:-1] + (lead[-1] * 32,))


def quantize_q8_0(w):
    """Format [EXT ggml k-quants, SURVEY.md App. C]; semantics fixed by the QMatMul::forward call sites src/openai/models/layers/attention.rs:920-922 and src/openai/models/quantized_llama.rs:33-37 (candle k_quants.rs is un-vendored)."""
    w = np.asarray(w, np.float32)
    lead = w.shape[:-1]
    x = w.reshape(lead + (w.shape[-1] // 32, 32))
    d = (np.abs(x).max(-1) / 127.0).astype(np.float16)
    df = d.astype(np.float32)[..., None]
    with np.errstate(divide="ignore", invalid="ignore"):
        q = np.where(df > 0, np.rint(x / df), 0).clip(-127, 127).astype(np.int8)
    out = np.zeros(lead + (x.shape[-2], Q8_0_BLOCK_BYTES), np.uint8)
    out[..., 0:2] = d[..., None].view(np.uint8)
    out[..., 2:34] = q.view(np.uint8)
    return out


def quantize_q8_0_ggml(w):
    """ggml `quantize_row_q8_0` [EXT] as candle's `BlockQ8_0::from_float` runs it when a tensor-parallel shard is re-quantised
    (layers/quantized_var_builder.rs:234-269): d = amax / 127 in f32, q = roundf(x * (1/d)) with the UNROUNDED d (half away
    from zero), the stored d rounded to f16."""
    w = np.asarray(w, np.float32)
    lead = w.shape[:-1]
    x = w.reshape(lead + (w.shape[-1] // 32, 32))
    d = (np.abs(x).max(-1) / np.float32(127.0)).astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        idv = np.where(d != 0, np.float32(1.0) / d, np.float32(0.0)).astype(np.float32)
    v = (x * idv[..., None]).astype(np.float32)
    q = (np.sign(v) * np.floor(np.abs(v) + np.float32(0.5))).astype(np.int8)          # roundf
    out = np.zeros(lead + (x.shape[-2], Q8_0_BLOCK_BYTES), np.uint8)
    out[..., 0:2] = d.astype(np.float16)[..., None].view(np.uint8)
    out[..., 2:34] = q.view(np.uint8)
    return out


def requantize_shard_q8_0(blocks, ggml_type, rank, world):
    """`get_sharded_no_shape`'s fallback (quantized_var_builder.rs:234-269) for a dim-1 shard that cuts a k-quant block:
    dequantize_f16 (f32 arithmetic, one rounding to f16) -> narrow to this rank's columns -> Q8_0.  blocks [N, K/256, bytes]."""
    w = dequantize(blocks, ggml_type).astype(np.float16).astype(np.float32)
    c = w.shape[-1] // world
    return quantize_q8_0_ggml(np.ascontiguousarray(w[:, rank * c:(rank + 1) * c]))


# --------------------------------------------------------------------------- Q8_K (CPU activation format, oracle O2)
def quantize_q8_k(x):
    """x: f32 [..., K] -> (d f32 [..., nb], q int8 [..., nb, 256], bsums int16 [..., nb, 16]).

    ggml quantize_row_q8_K: iscale = -128/max (signed max-abs element), q = min(127, round(iscale*x)),
    d = 1/iscale.  [EXT]
    Format [EXT ggml k-quants, SURVEY.md App. C]; semantics fixed by the QMatMul::forward call sites src/openai/models/layers/attention.rs:920-922 and src/openai/models/quantized_llama.rs:33-37 (candle k_quants.rs is un-vendored)."""
    x = np.asarray(x, np.float32)
    lead = x.shape[:-1]
    xb = x.reshape(lead + (x.shape[-1] // QK_K, QK_K))
    idx = np.abs(xb).argmax(-1)
    mx = np.take_along_axis(xb, idx[..., None], -1)[..., 0]
    with np.errstate(divide="ignore", invalid="ignore"):
        iscale = np.where(mx != 0, -128.0 / mx, 0).astype(np.float32)
        q = np.minimum(127, np.rint(iscale[..., None] * xb)).astype(np.int32)
        d = np.where(iscale != 0, 1.0 / iscale, 0).astype(np.float32)
    q = q.clip(-128, 127).astype(np.int8)
    bsums = q.reshape(lead + (xb.shape[-2], 16, 16)).astype(np.int32).sum(-1).astype(np.int16)
    return d, q, bsums


def vec_dot_q4k_q8k(blocks, xd, xq, xbsums):
    """O2: rows of Q4_K blocks [N, nb, 144] . one Q8_K-quantised vector -> f32 [N].
    Format [EXT ggml k-quants, SURVEY.md App. C]; semantics fixed by the QMatMul::forward call sites src/openai/models/layers/attention.rs:920-922 and src/openai/models/quantized_llama.rs:33-37 (candle k_quants.rs is un-vendored)."""
    b = np.ascontiguousarray(blocks, np.uint8)
    d = b[..., 0:2].copy().view(np.float16)[..., 0].astype(np.float32)
    dmin = b[..., 2:4].copy().view(np.float16)[..., 0].astype(np.float32)
    sc, m = _unpack_scales_q4k(b[..., 4:16])
    qs = b[..., 16:144].reshape(b.shape[:-1] + (4, 32))
    q = np.stack([qs & 0x0F, qs >> 4], axis=-2).reshape(b.shape[:-1] + (8, 32)).astype(np.int32)
    xq8 = xq.reshape(xq.shape[:-1] + (8, 32)).astype(np.int32)
    isum = (q * xq8[None]).sum(-1)                                 # [N, nb, 8]
    sumi = (isum * sc.astype(np.int32)).sum(-1)                    # [N, nb]
    bs = xbsums.reshape(xbsums.shape[:-1] + (8, 2)).astype(np.int32).sum(-1)   # per 32
    summ = (bs[None] * m.astype(np.int32)).sum(-1)
    acc = (xd[None] * d * sumi.astype(np.float32) - xd[None] * dmin * summ.astype(np.float32))
    return acc.sum(-1).astype(np.float32)


def vec_dot_q6k_q8k(blocks, xd, xq):
    """Format [EXT ggml k-quants, SURVEY.md App. C]; semantics fixed by the QMatMul::forward call sites src/openai/models/layers/attention.rs:920-922 and src/openai/models/quantized_llama.rs:33-37 (candle k_quants.rs is un-vendored)."""
    b = np.ascontiguousarray(blocks, np.uint8)
    lead = b.shape[:-1]
    ql = b[..., 0:128].reshape(lead + (2, 64))
    qh = b[..., 128:192].reshape(lead + (2, 32))
    sc = b[..., 192:208].copy().view(np.int8).reshape(lead + (16,)).astype(np.int32)
    d = b[..., 208:210].copy().view(np.float16)[..., 0].astype(np.float32)
    q1 = (ql[..., 0:32] & 0x0F) | (((qh >> 0) & 3) << 4)
    q2 = (ql[..., 32:64] & 0x0F) | (((qh >> 2) & 3) << 4)
    q3 = (ql[..., 0:32] >> 4) | (((qh >> 4) & 3) << 4)
    q4 = (ql[..., 32:64] >> 4) | (((qh >> 6) & 3) << 4)
    q = (np.stack([q1, q2, q3, q4], axis=-2).astype(np.int32) - 32).reshape(lead + (16, 16))
    xq16 = xq.reshape(xq.shape[:-1] + (16, 16)).astype(np.int32)
    isum = (q * xq16[None]).sum(-1) * sc
    return (xd[None] * d * isum.sum(-1).astype(np.float32)).sum(-1).astype(np.float32)


# --------------------------------------------------------------------------- mat-vec / mat-mat oracles
GGML_Q4_K = 12   # ggml type ids [EXT]
GGML_Q6_K = 14
GGML_Q8_0 = 8
BLOCK_BYTES = {GGML_Q4_K: Q4_K_BLOCK_BYTES, GGML_Q6_K: Q6_K_BLOCK_BYTES}


def dequantize(blocks, ggml_type):
    """Format [EXT ggml k-quants, SURVEY.md App. C]; semantics fixed by the QMatMul::forward call sites src/openai/models/layers/attention.rs:920-922 and src/openai/models/quantized_llama.rs:33-37 (candle k_quants.rs is un-vendored)."""
    if ggml_type == GGML_Q4_K:
        return dequantize_q4_k(blocks)
    if ggml_type == GGML_Q6_K:
        return dequantize_q6_k(blocks)
    if ggml_type == GGML_Q8_0:
        return dequantize_q8_0(blocks)
    raise ValueError(f"unsupported ggml type {ggml_type}")


def quantize(w, ggml_type):
    """Format [EXT ggml k-quants, SURVEY.md App. C]; semantics fixed by the QMatMul::forward call sites src/openai/models/layers/attention.rs:920-922 and src/openai/models/quantized_llama.rs:33-37 (candle k_quants.rs is un-vendored)."""
    if ggml_type == GGML_Q4_K:
        return quantize_q4_k(w)
    if ggml_type == GGML_Q6_K:
        return quantize_q6_k(w)
    if ggml_type == GGML_Q8_0:
        return quantize_q8_0(w)
    raise ValueError(f"unsupported ggml type {ggml_type}")


def qmatmul_o1(x, blocks, ggml_type):
    """`QMatMul::forward(&x_f32)` semantics, oracle O1: y[T,N] = x[T,K] . dequant(W[N,K])^T in f64.
    Format [EXT ggml k-quants, SURVEY.md App. C]; semantics fixed by the QMatMul::forward call sites src/openai/models/layers/attention.rs:920-922 and src/openai/models/quantized_llama.rs:33-37 (candle k_quants.rs is un-vendored)."""
    w = dequantize(blocks, ggml_type).astype(np.float64)           # [N, K]
    return (np.asarray(x, np.float64) @ w.T).astype(np.float32)


def qmatmul_o2(x, blocks, ggml_type):
    """candle-CPU-faithful (Q8_K activations, integer dot).
    Format [EXT ggml k-quants, SURVEY.md App. C]; semantics fixed by the QMatMul::forward call sites src/openai/models/layers/attention.rs:920-922 and src/openai/models/quantized_llama.rs:33-37 (candle k_quants.rs is un-vendored)."""
    x = np.asarray(x, np.float32)
    out = []
    for t in range(x.shape[0]):
        xd, xq, xb = quantize_q8_k(x[t])
        if ggml_type == GGML_Q4_K:
            out.append(vec_dot_q4k_q8k(blocks, xd, xq, xb))
        elif ggml_type == GGML_Q6_K:
            out.append(vec_dot_q6k_q8k(blocks, xd, xq))
        else:
            raise ValueError
    return np.stack(out)
