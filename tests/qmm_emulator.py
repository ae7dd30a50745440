"""Lane-accurate numpy emulation of candle_vllm_amd/csrc/qmatmul.hip's MFMA data path.

It consumes the REAL repacked tile bytes produced by `mi355_qweight_repack` (host C++ in the library) and
walks the same per-lane index formulas as `stage_x`, `load_tile`, `compute_q4k`, `compute_q6k` and the
hi+lo / C-layout epilogue, with the MFMA replaced by its documented semantics
(16x16xK: A lane l -> row l&15, k-group l>>4; B lane l -> col l&15, k-group l>>4;
 D lane l reg v -> row 4*(l>>4)+v, col l&15).  This pins the layout arithmetic on CPU; the GPU tests then
only have to catch HIP-level slips."""
import numpy as np

from oracle import ops

Q4K_TILE, Q6K_TILE = 2304, 3360
Q4_K, Q6_K = 12, 14


def bf16_bits(x):
    return ops.f32_to_bf16_bits(np.asarray(x, np.float32))


def bits_f32(b):
    return ops.bf16_bits_to_f32(np.asarray(b, np.uint16))


def f16(lo, hi):
    return np.array([lo | (hi << 8)], np.uint16).view(np.float16)[0].astype(np.float32)


def stage_x(x, BT):
    """x f32 [B, K] -> (ximg uint16 [K/8][2*BT][8], xs32 [K/32][2][BT], xs16 [K/16][2][BT])."""
    B, K = x.shape
    nE = K // 8
    ximg = np.zeros((nE, 2 * BT, 8), np.uint16)
    xs16 = np.zeros((K // 16, 2, BT), np.float32)
    xs32 = np.zeros((K // 32, 2, BT), np.float32)
    order = [0, 2, 1, 3, 4, 6, 5, 7]
    for b in range(B):
        v = x[b].reshape(nE, 8).astype(np.float32)
        hb = bf16_bits(v)
        hf = bits_f32(hb)
        lb = bf16_bits(v - hf)
        lf = bits_f32(lb)
        ximg[:, b, :] = hb[:, order]
        ximg[:, BT + b, :] = lb[:, order]
        xs16[:, 0, b] = hf.reshape(-1, 16).sum(1)
        xs16[:, 1, b] = lf.reshape(-1, 16).sum(1)
        xs32[:, 0, b] = hf.reshape(-1, 32).sum(1)
        xs32[:, 1, b] = lf.reshape(-1, 32).sum(1)
    return ximg, xs32, xs16


def _mfma(A, Bm):
    """A [64 lanes][kper] (row = l&15, kgroup = l>>4); Bm same for columns.  Returns D [64][4]."""
    kper = A.shape[1]
    Am = np.zeros((16, 4 * kper), np.float64)
    Bk = np.zeros((4 * kper, 16), np.float64)
    for l in range(64):
        Am[l & 15, (l >> 4) * kper:(l >> 4) * kper + kper] = A[l]
        Bk[(l >> 4) * kper:(l >> 4) * kper + kper, l & 15] = Bm[l]
    Dm = (Am @ Bk).astype(np.float32)
    D = np.zeros((64, 4), np.float32)
    for l in range(64):
        for v in range(4):
            D[l, v] = Dm[4 * (l >> 4) + v, l & 15]
    return D


def _u32(buf, off):
    return int(np.frombuffer(buf[off:off + 4].tobytes(), np.uint32)[0])


def _arow(m, BT):
    """a_row_of<BT>: rows beyond the staged batch re-read a staged row (their MFMA outputs are unused)"""
    mm = (m & 7) if (m & 7) < BT else BT - 1
    return True, (mm if m < 8 else BT + mm)


def compute_q4k(tile, ximg, xs32, kb, BT, NV, y):
    K128 = 0x43004300
    for p in range(2):
        for pr in range(2):
            for hi in range(2):
                j = 4 * p + 2 * pr + hi
                sh = hi * 4
                A = np.zeros((64, 8), np.float32)
                Bm = np.zeros((64, 8), np.float32)
                for l in range(64):
                    m, kg = l & 15, l >> 4
                    off = (256 if p == 0 else 1280) + l * 16
                    w0 = _u32(tile, off + 8 * pr)
                    w1 = _u32(tile, off + 8 * pr + 4)
                    words = [((w0 >> sh) & 0x000F000F) | K128, ((w0 >> (sh + 8)) & 0x000F000F) | K128,
                             ((w1 >> sh) & 0x000F000F) | K128, ((w1 >> (sh + 8)) & 0x000F000F) | K128]
                    bits = []
                    for w in words:
                        bits += [w & 0xFFFF, w >> 16]
                    Bm[l] = bits_f32(np.array(bits, np.uint16))
                    ok, arow = _arow(m, BT)
                    if ok:
                        E = (kb * 8 + j) * 4 + kg
                        A[l] = bits_f32(ximg[E, arow])
                D = _mfma(A, Bm)
                for l in range(64):
                    r, kg = l & 15, l >> 4
                    hdr = tile[r * 16: r * 16 + 16]
                    d = f16(int(hdr[0]), int(hdr[1]))
                    dmin = f16(int(hdr[2]), int(hdr[3]))
                    s = hdr[4:16].astype(np.int64)
                    if j < 4:
                        sc, mn = s[j] & 63, s[j + 4] & 63
                    else:
                        sc = (s[j + 4] & 0xF) | ((s[j - 4] >> 6) << 4)
                        mn = (s[j + 4] >> 4) | ((s[j] >> 6) << 4)
                    dsc = np.float32(d * sc)
                    cj = np.float32(dmin * mn + 128.0 * d * sc)
                    hl = kg >> 1
                    for v in range(NV):
                        xsum = xs32[kb * 8 + j, hl, (4 * (kg & 1) + v) & (BT - 1)]
                        y[l, v] += dsc * D[l, v] - cj * xsum


def compute_q6k(tile, ximg, xs16, kb, BT, NV, y):
    K128 = 0x43004300
    for n in range(2):
        for is_ in range(2):
            for tt in range(4):
                s = 8 * n + 2 * tt + is_
                A = np.zeros((64, 4), np.float32)
                Bm = np.zeros((64, 4), np.float32)
                for l in range(64):
                    m, kg = l & 15, l >> 4
                    qoff = (256 if n == 0 else 1280) + l * 16
                    a = _u32(tile, qoff + 8 * is_)
                    b = _u32(tile, qoff + 8 * is_ + 4)
                    h = _u32(tile, 2304 + l * 16 + 4 * (2 * n + is_))
                    t = [(a & 0x0F0F0F0F) | ((h << 4) & 0x30303030),
                         (b & 0x0F0F0F0F) | ((h << 2) & 0x30303030),
                         ((a >> 4) & 0x0F0F0F0F) | (h & 0x30303030),
                         ((b >> 4) & 0x0F0F0F0F) | ((h >> 2) & 0x30303030)][tt]
                    words = [(t & 0x00FF00FF) | K128, ((t >> 8) & 0x00FF00FF) | K128]
                    bits = []
                    for w in words:
                        bits += [w & 0xFFFF, w >> 16]
                    Bm[l] = bits_f32(np.array(bits, np.uint16))
                    ok, arow = _arow(m, BT)
                    if ok:
                        E = kb * 32 + 2 * s + (kg >> 1)
                        half = kg & 1
                        A[l] = bits_f32(ximg[E, arow, 4 * half: 4 * half + 4])
                D = _mfma(A, Bm)
                for l in range(64):
                    r, kg = l & 15, l >> 4
                    sc8 = int(np.int8(tile[r * 16 + s]))
                    d = f16(int(tile[3328 + 2 * r]), int(tile[3328 + 2 * r + 1]))
                    dsc = np.float32(d * sc8)
                    hl = kg >> 1
                    for v in range(NV):
                        xsum = xs16[kb * 16 + s, hl, (4 * (kg & 1) + v) & (BT - 1)]
                        y[l, v] += dsc * (D[l, v] - np.float32(160.0) * xsum)


def emulate_qmatmul(x, tiles, ggml_type, N, K):
    """x f32 [B<=8, K]; tiles: uint8 repacked buffer.  Returns y f32 [B, N] as the kernel computes it."""
    B = x.shape[0]
    BT = 1 if B == 1 else 2 if B == 2 else 4 if B <= 4 else 8
    NV = min(BT, 4)
    nkb = K // 256
    tb = Q4K_TILE if ggml_type == Q4_K else Q6K_TILE
    ximg, xs32, xs16 = stage_x(x, BT)
    out = np.zeros((B, N), np.float32)
    for rt in range((N + 15) // 16):
        y = np.zeros((64, NV), np.float32)
        for kb in range(nkb):
            tile = tiles[(rt * nkb + kb) * tb:(rt * nkb + kb + 1) * tb]
            if ggml_type == Q4_K:
                compute_q4k(tile, ximg, xs32, kb, BT, NV, y)
            else:
                compute_q6k(tile, ximg, xs16, kb, BT, NV, y)
        for l in range(32):                      # y[v] += shfl_xor(y[v], 32); lanes kg<2 hold batch 4kg+v
            kg, r = l >> 4, l & 15
            for v in range(NV):
                b = 4 * kg + v
                row = rt * 16 + r
                if b < B and row < N:
                    out[b, row] = y[l, v] + y[l + 32, v]
    return out
