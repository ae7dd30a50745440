"""The permutations of the LDS-DMA decode-attention experiment (paged_attn_lds_kernel, tuning key 44 = 2), checked on the host: the kernel
and this test call the SAME index functions (csrc/paged_attention.hip: pal_*; here through mi355_internal_pal_layout).  A 64-token stage
is written into a byte image exactly as the 32 DMA pieces write it (lane-linear destination, permuted per-lane source), then every lane's
fragment reads are taken from the image and compared with what the MFMA conventions of the kernel need:
  * K tile (ip, it), A-operand row r, k-group kg, step j  ->  token 32 ip + 8 (r >> 2) + (r & 3) + 4 it, channels 32 j + 8 kg .. + 7
    (score row 4 kg + v of the tile is then token 32 ip + 8 kg + 4 it + v: the mask / softmax indexing of the kernel);
  * V pair ip, B-operand column = channel, k-group kg        ->  tokens 32 ip + 8 kg .. + 7 of that channel
    (= the probabilities the lane packs into its A fragment: tile A rows 4 kg + v, tile B rows 4 kg + v);
and the b128 lane groups of both reads are bank-conflict free (MI355X_MICROARCH.md: 64 banks of 4 B, groups of 16 lanes)."""
import ctypes

import numpy as np
import pytest


@pytest.fixture(scope="module")
def pal():
    import __graft_entry__ as ge
    ge.build()
    from candle_vllm_amd._lib import lib
    f = lib.mi355_internal_pal_layout
    f.restype = ctypes.c_int32
    f.argtypes = [ctypes.c_int32] * 6
    return lambda what, a=0, b=0, c=0, d=0, e=0: int(f(what, a, b, c, d, e))


def _stage_image(pal):
    """uint16 image [16384] of one stage; element value: K -> token * 128 + channel, V -> 0x4000 | (token * 128 + channel)"""
    img = np.full(16384, 0xFFFF, np.uint16)
    written = np.zeros(16384, bool)
    for q in range(32):
        for lane in range(64):
            row, tok = pal(0, q, lane), pal(1, q, lane)
            dst = (q * 1024 + lane * 16) // 2
            if q < 16:        # 16 bytes = the 8 channels of channel group `row` of token `tok`
                vals = [tok * 128 + 8 * row + e for e in range(8)]
            else:             # 16 bytes = tokens tok .. tok + 7 of channel `row`
                assert tok % 8 == 0 and 0 <= tok < 64 and 0 <= row < 128
                vals = [0x4000 | ((tok + e) * 128 + row) for e in range(8)]
            assert not written[dst:dst + 8].any()
            img[dst:dst + 8] = vals
            written[dst:dst + 8] = True
    assert written.all()
    return img


def test_every_granule_of_the_stage_is_copied_exactly_once(pal):
    img = _stage_image(pal)
    k, v = img[:8192], img[8192:]
    assert sorted(k.tolist()) == list(range(64 * 128))                       # every (token, channel) of K once
    assert sorted((v & 0x3FFF).tolist()) == list(range(64 * 128)) and (v & 0x4000).all()
    # the sources of one piece stay inside one 1 KiB K row / eight 128-byte V rows: full lines on the global side
    for q in range(32):
        rows = {pal(0, q, lane) for lane in range(64)}
        assert len(rows) == (1 if q < 16 else 8)


def test_fragment_reads_find_the_operands_the_mfma_conventions_need(pal):
    img = _stage_image(pal)
    for ip in range(2):
        for it in range(2):
            for j in range(4):
                for kg in range(4):
                    for r in range(16):
                        off = pal(2, j, kg, ip, it, r)
                        assert off % 16 == 0
                        got = img[off // 2: off // 2 + 8]
                        tok = 32 * ip + 8 * (r >> 2) + (r & 3) + 4 * it
                        assert got.tolist() == [tok * 128 + 32 * j + 8 * kg + e for e in range(8)]
    for ch in range(128):
        for ip in range(2):
            for kg in range(4):
                off = pal(3, ch, ip, kg)
                got = img[off // 2: off // 2 + 8]
                assert got.tolist() == [0x4000 | ((32 * ip + 8 * kg + e) * 128 + ch) for e in range(8)]
    # the probabilities a lane packs: tile A row 4 kg + v -> token 32 ip + 8 kg + v, tile B -> + 4: together tokens 8 kg .. 8 kg + 7
    for kg in range(4):
        rows_a = [32 * 0 + 8 * ((4 * kg + v) >> 2) + ((4 * kg + v) & 3) for v in range(4)]
        assert rows_a == [8 * kg + v for v in range(4)]


def test_b128_lane_groups_are_bank_conflict_free(pal):
    groups = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
              list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)), list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]

    def quads(offs):            # the 4-bank quads (16 B) a group touches inside the 256-byte bank row
        return sorted((o // 16) % 16 for o in offs)

    for grp in groups:
        for ip in range(2):
            for it in range(2):
                for j in range(4):
                    offs = [pal(2, j, lane >> 4, ip, it, lane & 15) for lane in grp]
                    assert quads(offs) == list(range(16)), ("K", ip, it, j)
            for wave in range(4):
                for n2 in range(2):
                    offs = [pal(3, 32 * wave + 16 * n2 + (lane & 15), ip, lane >> 4) for lane in grp]
                    assert quads(offs) == list(range(16)), ("V", ip, wave, n2)


def test_stream_cuts_cover_every_stage_once_and_the_reduce_finds_the_workgroups(pal):
    """paged_attn_stream_kernel: the (sequence, 64-token stage) pairs of a kv head form one flat index space, workgroup w takes
    [cut(w), cut(w + 1)); the reduce kernel decides with the same cuts which workgroups left a partial for its sequence"""
    rng = np.random.default_rng(7)
    for trial in range(60):
        B = int(rng.integers(1, 65))
        W = int(rng.choice([1, 2, 7, 16, 32, 64]))
        ctx = rng.integers(1, [65, 300, 4097][trial % 3], B)
        n = (ctx + 63) // 64
        pend = np.cumsum(n)
        S = int(pend[-1])
        cuts = [pal(4, S, W, w) for w in range(W + 1)]
        assert cuts[0] == 0 and cuts[-1] == S and all(0 <= cuts[w + 1] - cuts[w] <= -(-S // W) for w in range(W))
        assert max(cuts[w + 1] - cuts[w] for w in range(W)) - min(cuts[w + 1] - cuts[w] for w in range(W)) <= 1
        owner = np.full(S, -1)
        wrote = set()                                                  # (sequence, workgroup) partials
        for w in range(W):
            for f in range(cuts[w], cuts[w + 1]):
                assert owner[f] == -1
                owner[f] = w
                b = int(np.searchsorted(pend, f, side="right"))       # = popcount(pend <= f)
                wrote.add((b, w))
        assert (owner >= 0).all()
        for b in range(B):
            ps, pe = int(pend[b] - n[b]), int(pend[b])
            meets = {w for w in range(W) if cuts[w] < cuts[w + 1] and cuts[w] < pe and cuts[w + 1] > ps}
            assert meets == {w for (bb, w) in wrote if bb == b} and meets



def _stage_image_fp8(pal):
    """uint16 image [16384] of one e4m3fn stage, one element per BYTE; value: K -> token * 128 + channel, V -> 0x4000 | (token * 128 + channel)"""
    img = np.full(16384, 0xFFFF, np.uint16)
    written = np.zeros(16384, bool)
    for q in range(16):
        for lane in range(64):
            row, tok = pal(5, q, lane), pal(6, q, lane)
            dst = q * 1024 + lane * 16
            if q < 8:         # 16 bytes = the 16 channels of channel group `row` of token `tok`
                vals = [tok * 128 + 16 * row + e for e in range(16)]
            else:             # 16 bytes = tokens tok .. tok + 15 of channel `row`
                assert tok % 16 == 0 and 0 <= tok < 64 and 0 <= row < 128
                vals = [0x4000 | ((tok + e) * 128 + row) for e in range(16)]
            assert not written[dst:dst + 16].any()
            img[dst:dst + 16] = vals
            written[dst:dst + 16] = True
    assert written.all()
    return img


def test_fp8_stage_every_granule_once_and_the_fragments_the_token_split_needs(pal):
    """round 5, paged_attn_stream_kernel<.., KV8>: the 16 KiB stage of the e4m3fn cache -- written as the 16 DMA pieces write it, read back as
    the lanes of the token split read it: K tile (ip, it), row r, k-group kg, step j -> 8 bytes = channels 32 j + 8 kg .. + 7 of token
    32 ip + 8 (r >> 2) + (r & 3) + 4 it (the bf16 stage's token order: masks and softmax indexing are shared); V channel ch, tile (ip, it),
    k-group kg -> 4 bytes = tokens 32 ip + 8 kg + 4 it .. + 3 (the lane's four probabilities)."""
    img = _stage_image_fp8(pal)
    k, v = img[:8192], img[8192:]
    assert sorted(k.tolist()) == list(range(64 * 128))
    assert sorted((v & 0x3FFF).tolist()) == list(range(64 * 128)) and (v & 0x4000).all()
    for q in range(16):                                               # full lines on the global side: one K row / sixteen 64-byte V rows
        assert len({pal(5, q, lane) for lane in range(64)}) == (1 if q < 8 else 16)
    for ip in range(2):
        for it in range(2):
            for j in range(4):
                for kg in range(4):
                    for r in range(16):
                        off = pal(7, j, kg, ip, it, r)
                        assert off % 8 == 0
                        tok = 32 * ip + 8 * (r >> 2) + (r & 3) + 4 * it
                        assert img[off:off + 8].tolist() == [tok * 128 + 32 * j + 8 * kg + e for e in range(8)]
            for ch in range(128):
                for kg in range(4):
                    off = pal(8, ch, ip, kg, it)
                    assert off % 4 == 0
                    assert img[off:off + 4].tolist() == [0x4000 | ((32 * ip + 8 * kg + 4 * it + e) * 128 + ch) for e in range(4)]


def test_fp8_stage_fragment_reads_are_bank_conflict_free(pal):
    """64 banks of 4 B.  The 4-byte V reads of a wave instruction (64 lanes: channel 16 n2 + (lane & 15), k-group lane >> 4) all carry the
    wave's own `it`, which selects the even or the odd banks of a 16-byte slot: 32 banks at best, and the swizzle reaches that floor (every
    bank of the 32 exactly twice -- two passes, what the 8-byte reads of the bf16 stage take as well; without the swizzle it is 8-way).
    The 8-byte K reads are served 32 lanes at a time and each half covers 32 different bank pairs."""
    for ip in range(2):
        for it in range(2):
            for n2 in range(8):
                banks = [(pal(8, 16 * n2 + (lane & 15), ip, lane >> 4, it) // 4) % 64 for lane in range(64)]
                assert sorted(banks) == sorted(2 * [b for b in range(64) if b % 2 == it % 2]), ("V", ip, it, n2)
            for j in range(4):
                for half in range(2):
                    pairs = [(pal(7, j, lane >> 4, ip, it, lane & 15) // 8) % 32 for lane in range(32 * half, 32 * half + 32)]
                    assert sorted(pairs) == list(range(32)), ("K", ip, it, j, half)
