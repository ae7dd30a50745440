"""Lane-level emulation of the LDS-DMA prompt-attention experiment (csrc/prefill_attention.hip: prefill_attn_lds_kernel, tuning key 47)
on the host: the kernel was written after the round's GPU minutes were spent, so its DATA FLOW is checked here -- the stage image as the
DMA pieces write it (through the kernel's own index functions, mi355_internal_pal_layout), the fragment reads, the 16x16x32 MFMA operand
conventions (A: lane = row l & 15, k = 8 (l >> 4) + e; B: lane = column l & 15, same k; C / D: lane = column l & 15, rows 4 (l >> 4) + v),
the per-lane online softmax over query tiles, the causal masks, P as hi + lo bf16 A operands, the alpha broadcast and the output
addressing -- against the oracle's prefill attention.  The emulation follows the kernel statement by statement; it proves the design, not
the C++ text (that is test_gpu_prefill.py -k lds_dma on the GPU)."""
import ctypes

import numpy as np
import pytest

from oracle import ops as O

LANES = np.arange(64)
C_, KG = LANES & 15, LANES >> 4


@pytest.fixture(scope="module")
def pal():
    import __graft_entry__ as ge
    ge.build()
    from candle_vllm_amd._lib import lib
    f = lib.mi355_internal_pal_layout
    f.restype = ctypes.c_int32
    f.argtypes = [ctypes.c_int32] * 6
    return lambda what, a=0, b=0, c=0, d=0, e=0: int(f(what, a, b, c, d, e))


def mfma(a, b, c):
    """v_mfma_f32_16x16x32: a, b [64 lanes][8], c [64][4] -> d [64][4]"""
    A = np.zeros((16, 32), np.float64)
    B = np.zeros((32, 16), np.float64)
    for l in range(64):
        A[l & 15, 8 * (l >> 4): 8 * (l >> 4) + 8] = a[l]
        B[8 * (l >> 4): 8 * (l >> 4) + 8, l & 15] = b[l]
    Dm = A @ B
    d = c.astype(np.float64).copy()
    for l in range(64):
        d[l] += Dm[4 * (l >> 4): 4 * (l >> 4) + 4, l & 15]
    return d.astype(np.float32)


def xor_reduce(x, op):
    x = op(x, x[LANES ^ 16])
    return op(x, x[LANES ^ 32])


def emulate_workgroup(pal, q, k, v, cached, q0, hk, G, hg, scale, out):
    """one workgroup: queries q0 .. q0 + 63 of the heads hk * G + 4 hg + wave; k, v [ctx, Hkv, 128] (what the cache holds)"""
    qlen, H, D = q.shape
    ctx = cached + qlen
    kend = min(ctx, cached + min(q0 + 64, qlen))
    ns = (kend + 63) >> 6
    scale_log2 = np.float32(scale * 1.4426950408889634)
    state = {}
    for wave in range(4):
        hw = 4 * hg + wave
        if hw >= G:
            continue
        h = hk * G + hw
        qf = np.zeros((4, 4, 64, 8), np.float32)
        pos = np.zeros((4, 64), np.int64)
        for qt in range(4):
            ql = q0 + 16 * qt + C_
            pos[qt] = cached + np.minimum(ql, qlen - 1)
            for j in range(4):
                for l in range(64):
                    if ql[l] < qlen:
                        qf[qt, j, l] = q[ql[l], h, 32 * j + 8 * KG[l]: 32 * j + 8 * KG[l] + 8]
        state[wave] = dict(h=h, qf=qf, pos=pos, m=np.full((4, 64), -np.inf, np.float32), l=np.zeros((4, 64), np.float32),
                           o=np.zeros((4, 8, 64, 4), np.float32))
    for i in range(ns):
        # ---- the stage image, as the 32 DMA pieces write it (tokens at or beyond ctx: the piece reads the stage's token 0 instead;
        # what lies beyond ctx inside a fetched V granule is whatever the cache holds: NaN here)
        img = np.full(16384, np.nan, np.float32)
        for piece in range(32):
            for lane in range(64):
                row, tok = pal(0, piece, lane), pal(1, piece, lane)
                if 64 * i + tok >= ctx:
                    tok = 0
                dst = (piece * 1024 + lane * 16) // 2
                if piece < 16:
                    img[dst: dst + 8] = k[64 * i + tok, hk, 8 * row: 8 * row + 8]
                else:
                    for e in range(8):
                        t = 64 * i + tok + e
                        img[dst + e] = v[t, hk, row] if t < ctx else np.nan
        tb = 64 * i
        diag = tb + 63 > cached + q0
        for wave, stt in state.items():
            ka = np.zeros((2, 2, 4, 64, 8), np.float32)
            for ip in range(2):
                for it in range(2):
                    for j in range(4):
                        for l in range(64):
                            off = pal(2, j, int(KG[l]), ip, it, int(C_[l])) // 2
                            ka[ip, it, j, l] = img[off: off + 8]
            pb = np.zeros((4, 2, 64, 8), np.float32)
            pl = np.zeros((4, 2, 64, 8), np.float32)
            alpha = np.zeros((4, 64), np.float32)
            for qt in range(4):
                x = np.zeros((2, 2, 64, 4), np.float32)
                for ip in range(2):
                    for it in range(2):
                        acc = np.zeros((64, 4), np.float32)
                        for j in range(4):
                            acc = mfma(ka[ip, it, j], stt["qf"][qt, j], acc)
                        x[ip, it] = acc * scale_log2
                        if diag:
                            for vv in range(4):
                                tok = tb + 32 * ip + 8 * KG + 4 * it + vv
                                x[ip, it, :, vv] = np.where(tok <= stt["pos"][qt], x[ip, it, :, vv], -np.inf)
                tmax = xor_reduce(x.transpose(2, 0, 1, 3).reshape(64, -1).max(1), np.maximum)
                m_new = np.maximum(stt["m"][qt], tmax)
                assert np.isfinite(m_new).all()
                alpha[qt] = np.exp2(stt["m"][qt] - m_new)
                stt["m"][qt] = m_new
                psum = np.zeros(64, np.float32)
                for ip in range(2):
                    e8 = np.concatenate([np.exp2(x[ip, 0] - m_new[:, None]), np.exp2(x[ip, 1] - m_new[:, None])], axis=1).astype(np.float32)
                    psum += e8.sum(1)
                    hi = O.round_bf16(e8)
                    pb[qt, ip] = hi
                    pl[qt, ip] = O.round_bf16(e8 - hi)
                stt["l"][qt] = stt["l"][qt] * alpha[qt] + psum
            for nt in range(8):
                vfr = np.zeros((2, 64, 8), np.float32)
                for ip in range(2):
                    for l in range(64):
                        off = pal(3, 16 * nt + int(C_[l]), ip, int(KG[l])) // 2
                        vals = img[off: off + 8].copy()
                        tk = tb + 32 * ip + 8 * KG[l]
                        vals[tk + np.arange(8) >= ctx] = 0.0                     # the kernel's bit mask on the B fragment
                        vfr[ip, l] = vals
                for qt in range(4):
                    on = stt["o"][qt, nt].copy()
                    for vv in range(4):
                        on[:, vv] *= alpha[qt][4 * KG + vv]                       # ds_bpermute from lane 4 kg + v
                    for ip in range(2):
                        on = mfma(pb[qt, ip], vfr[ip], on)
                        on = mfma(pl[qt, ip], vfr[ip], on)
                    stt["o"][qt, nt] = on
    for wave, stt in state.items():
        for qt in range(4):
            lt = xor_reduce(stt["l"][qt], np.add)
            for vv in range(4):
                lq = lt[4 * KG + vv]
                ql = q0 + 16 * qt + 4 * KG + vv
                for l in range(64):
                    if ql[l] < qlen:
                        for nt in range(8):
                            out[ql[l], stt["h"], 16 * nt + C_[l]] = stt["o"][qt, nt, l, vv] / lq[l]


@pytest.mark.parametrize("qlen,cached,H,Hkv", [(70, 50, 4, 2), (64, 0, 7, 1), (5, 130, 4, 4)])
def test_prefill_lds_data_flow_equals_the_oracle(pal, qlen, cached, H, Hkv):
    rng = np.random.default_rng(qlen + cached)
    D, G = 128, H // Hkv
    ctx = cached + qlen
    q = O.round_bf16(rng.normal(0, 1, (qlen, H, D)).astype(np.float32))
    k = O.round_bf16(rng.normal(0, 1, (ctx, Hkv, D)).astype(np.float32))
    v = O.round_bf16(rng.normal(0, 1, (ctx, Hkv, D)).astype(np.float32))
    scale = 1.0 / np.sqrt(D)
    out = np.full((qlen, H, D), np.nan, np.float32)
    for qb in range((qlen + 63) // 64):
        for hk in range(Hkv):
            for hg in range((G + 3) // 4):
                emulate_workgroup(pal, q, k, v, cached, 64 * qb, hk, G, hg, scale, out)
    assert np.isfinite(out).all()                                                 # every (query, head, channel) written, no NaN leaked
    ref = O.prefill_attention(q, k, v, scale, cached=cached, rnd=lambda a: a)
    assert np.abs(out - ref).max() <= 5e-5 * max(1.0, np.abs(ref).max())        # hi + lo probabilities, f32 accumulation (measured 2e-6 .. 7e-6)
