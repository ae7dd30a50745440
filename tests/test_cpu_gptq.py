"""CPU tests for the safetensors-path oracle (GPTQ/AWQ/Marlin tensors, 16-bit linears): the oracle against the
golden vectors generated from the reference's in-tree Python, and the C-ABI host helpers against the oracle."""
import json
import os

import numpy as np
import pytest

from oracle import gptq as G

GOLD = os.path.join(os.path.dirname(__file__), "golden", "marlin_perms.json")


@pytest.fixture(scope="module")
def gold():
    with open(GOLD) as f:
        return json.load(f)


def test_scale_perms_match_reference(gold):
    sp, sps = G.scale_perms()
    assert sp.tolist() == gold["scale_perm"]
    assert sps.tolist() == gold["scale_perm_single"]


def test_pack_and_zero_points_match_reference(gold):
    for c in gold["cases"]:
        zp = np.array(c["zp"], np.int64)
        Gn, N = c["G"], c["N"]
        assert (G.pack_cols(zp) == np.array(c["pack_cols"], np.uint32)).all()
        assert (G.unpack_cols(np.array(c["pack_cols"], np.uint32)) == zp).all()
        assert (G.marlin_zero_points(zp, Gn, N) == np.array(c["marlin_zero_points"], np.uint32)).all()
        awq = np.array(c["awq_packed"], np.uint32)
        assert (G.awq_pack(zp) == awq).all()                     # AWQ nibble order
        assert (G.awq_unpack(awq) == zp).all()
        assert (G.awq_to_marlin_zero_points(awq, Gn, N) == np.array(c["awq_to_marlin_zero_points"], np.uint32)).all()


def test_gptq_pack_roundtrip():
    rng = np.random.default_rng(0)
    q = rng.integers(0, 16, (256, 48))
    assert (G.gptq_unpack(G.gptq_pack(q)) == q).all()
    w = G.gptq_pack(q)
    assert ((w[3, 5] >> np.uint32(4 * 6)) & 0xF) == q[3 * 8 + 6, 5]


def test_host_index_helpers_invert_the_permutations(lib, gold):
    """mi355_marlin_scale_pos / zero_pos (the kernels' index arithmetic) vs the reference permutations."""
    N = 256
    nat = np.arange(N)
    for grouped, gs in ((1, 128), (0, -1)):
        perm = G.marlin_permute_scales(nat.reshape(1, N).astype(np.float32), 1024, N, gs)[0].astype(np.int64)
        for n in range(N):
            assert perm[lib.mi355_marlin_scale_pos(n, grouped)] == n
    for c in gold["cases"]:
        zp, N = np.array(c["zp"]), c["N"]
        packed = np.array(c["marlin_zero_points"], np.uint32)
        for g in range(c["G"]):
            for n in range(N):
                p = lib.mi355_marlin_zero_pos(n)
                assert ((packed[g, p // 8] >> np.uint32(4 * (p % 8))) & 0xF) == zp[g, n]


def test_marlin_format_weight_permutation(lib):
    """the Marlin-format weight permutation: the library's closed form == the restated `_get_perms`; it is a permutation of
    1024; pack -> (host statement of the device un-permute) recovers the GPTQ words bit for bit"""
    perm = G.marlin_weight_perm()
    assert sorted(perm.tolist()) == list(range(1024))
    assert [lib.mi355_marlin_weight_perm(j) for j in range(1024)] == perm.tolist()
    assert lib.mi355_marlin_weight_perm(-1) == -1 and lib.mi355_marlin_weight_perm(1024) == -1
    # first entries by hand (i = 0: col 0, rows 0,1,8,9 of block 0 then block 1; interleave [0,2,4,6,1,3,5,7])
    assert perm[:8].tolist() == [0, 128, 8, 136, 16, 144, 24, 152]
    rng = np.random.default_rng(2)
    K, N = 64, 128
    q = rng.integers(0, 16, (K, N))
    B = G.marlin_format_pack(q)
    assert B.shape == (K // 16, 2 * N)
    inv = np.argsort(perm)
    out = np.zeros((K // 8, N), np.uint32)
    for kr in range(K // 8):
        for n in range(N):
            for i in range(8):
                k = 8 * kr + i
                src = (n >> 4) * 256 + (k & 15) * 16 + (n & 15)
                col = (src & ~1023) + int(inv[src & 1023])
                out[kr, n] |= ((B[k >> 4, col >> 3] >> np.uint32(4 * (col & 7))) & np.uint32(0xF)) << np.uint32(4 * i)
    assert (out == G.gptq_pack(q)).all()


def test_dequant_and_linear_definitions():
    rng = np.random.default_rng(1)
    K, N, gs = 256, 32, 64
    q = rng.integers(0, 16, (K, N))
    s = G.round_dt(rng.uniform(0.01, 0.03, (K // gs, N)), "f16")
    w = G.gptq_dequant(q, s, None, gs)
    assert w.shape == (K, N)
    assert np.allclose(w[70, 3], s[1, 3] * (q[70, 3] - 8))
    z = rng.integers(1, 17, (K // gs, N))
    w2 = G.gptq_dequant(q, s, z, gs)
    assert np.allclose(w2[130, 7], s[2, 7] * (q[130, 7] - z[2, 7]))
    gi = rng.permutation(K) // gs
    w3 = G.gptq_dequant(q, s, z, g_idx=gi)
    assert np.allclose(w3[5, 1], s[gi[5], 1] * (q[5, 1] - z[gi[5], 1]))
    assert (G.unpack_cols(G.gptq_pack_zeros(z)) + 1 == z).all()
    x = G.round_dt(rng.normal(0, 1, (3, K)), "bf16")
    wd = G.round_dt(rng.normal(0, 0.05, (N, K)), "bf16")
    b = G.round_dt(rng.normal(0, 0.1, N), "bf16")
    y = G.linear16(x, wd, b, "bf16")
    assert (G.round_dt(y, "bf16") == y).all()
    ref = (x.astype(np.float64) @ wd.astype(np.float64).T) + b
    assert np.abs(y - ref).max() <= 2 ** -7 * np.abs(ref).max()


def test_gptq_tile_order_host_statement(lib):
    """the 16-column x 256-k tile order of the marlin-slot weight image: the library's index function == the numpy statement
    (oracle/gptq.py gptq_tile), a permutation of the words; shapes the marlin arm refuses are not tiled"""
    rng = np.random.default_rng(9)
    K, N = 512, 48
    qw = rng.integers(0, 2 ** 32, (K // 8, N), dtype=np.uint32)
    tiled = G.gptq_tile(qw)
    idx = np.array([[lib.mi355_gptq_tile_index(kr, n, K, N) for n in range(N)] for kr in range(K // 8)])
    assert sorted(idx.reshape(-1).tolist()) == list(range(qw.size))
    assert (tiled[idx] == qw).all()
    # one wave's share of a k-block: lane l of half h holds 4 consecutive words = rows 32 kb + 4 (4h + jj) + l / 16, column 16 tile + l % 16
    assert lib.mi355_gptq_tile_index(0, 0, K, N) == 0 and lib.mi355_gptq_tile_index(4, 0, K, N) == 1
    assert lib.mi355_gptq_tile_index(1, 0, K, N) == 16 * 4 and lib.mi355_gptq_tile_index(16, 0, K, N) == 256
    assert lib.mi355_gptq_tile_index(32, 0, K, N) == 512 and lib.mi355_gptq_tile_index(0, 16, K, N) == 512 * (K // 256)
    assert lib.mi355_gptq_tile_index(0, 0, 128, N) == -1 and lib.mi355_gptq_tile_index(0, 0, K, 40) == -1
    assert lib.mi355_gptq_tile_index(K // 8, 0, K, N) == -1 and lib.mi355_gptq_tile_index(0, N, K, N) == -1


def test_dense_tile_order_host_statement(lib):
    """the 16-row x 256-k tile order of the 16-bit host layer's projections: the library's index function == the numpy
    statement (oracle/gptq.py dense_tile), a permutation of the elements; other shapes are not tiled"""
    N, K = 48, 512
    w = np.arange(N * K, dtype=np.uint32).reshape(N, K)
    tiled = G.dense_tile(w)
    rng = np.random.default_rng(3)
    for n, k in [(0, 0), (0, 7), (0, 8), (1, 0), (0, 32), (0, 256), (16, 0), (47, 511)] + [(int(a), int(b)) for a, b in zip(rng.integers(0, N, 200), rng.integers(0, K, 200))]:
        assert tiled[lib.mi355_dense_tile_index(n, k, N, K)] == w[n, k]
    assert sorted(tiled.tolist()) == list(range(N * K))
    # fragment j of a (tile, k-block): 64 lanes x 8 elements contiguous; lane = 16 * k-group + row
    assert lib.mi355_dense_tile_index(1, 0, N, K) == 8 and lib.mi355_dense_tile_index(0, 8, N, K) == 128
    assert lib.mi355_dense_tile_index(0, 32, N, K) == 512 and lib.mi355_dense_tile_index(0, 256, N, K) == 4096
    assert lib.mi355_dense_tile_index(16, 0, N, K) == 4096 * (K // 256)
    assert lib.mi355_dense_tile_index(0, 0, 40, K) == -1 and lib.mi355_dense_tile_index(0, 0, N, 128) == -1
    assert lib.mi355_dense_tile_index(N, 0, N, K) == -1 and lib.mi355_dense_tile_index(0, K, N, K) == -1
