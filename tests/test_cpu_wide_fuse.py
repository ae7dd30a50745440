"""Host-side check of the arrival arithmetic of the wide path's in-launch split-K epilogue (csrc/qmm_wide1_gemm.inc): a workgroup takes
one ticket per 256-column block of the output its row tiles fall into, and the block's epilogue runs in the workgroup whose ticket equals
k-splits x (workgroups touching the block) - 1.  `qw1_wgs_touching` -- the same function the kernels call -- must equal a brute-force
count for every run layout the launcher can produce (one launch, the merged Q4_K + Q6_K launch, several launches), or a block would be
finished early (missing partial sums) or never."""
import ctypes

import numpy as np
import pytest

NC = 7                                                                # QMG_NC: consumer waves = row tiles per workgroup


def _f(lib):
    f = lib.mi355_internal_qw1_wgs_touching                          # internal symbol (not in the public header): typed here
    f.restype = ctypes.c_int32
    f.argtypes = [ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32]
    return f


def _brute(runs, n_blocks):
    touch = np.zeros(n_blocks, np.int64)
    for lo, hi in runs:
        for wg_lo in range(lo, hi, NC):
            tiles = range(wg_lo, min(hi, wg_lo + NC))
            for blk in sorted({t >> 4 for t in tiles}):
                touch[blk] += 1
    return touch


@pytest.mark.parametrize("seed", range(8))
def test_ticket_expectation_equals_brute_force(lib, seed):
    rng = np.random.default_rng(seed)
    for _ in range(200):
        n_runs = int(rng.integers(1, 4))
        sizes = [int(rng.integers(1, 400)) for _ in range(n_runs)]
        runs, base = [], 0
        for s in sizes:
            runs.append((base, base + s))
            base += s
        n_blocks = (base * 16 + 255) // 256
        want = _brute(runs, n_blocks)
        lo = (ctypes.c_int32 * 3)(*[r[0] for r in runs] + [0] * (3 - n_runs))
        hi = (ctypes.c_int32 * 3)(*[r[1] for r in runs] + [0] * (3 - n_runs))
        got = [_f(lib)(n_runs, ctypes.addressof(lo), ctypes.addressof(hi), b) for b in range(n_blocks)]
        assert got == want.tolist(), (runs, got, want.tolist())
        # every workgroup's own block range [tile_lo >> 4, (tile_hi - 1) >> 4] is what the brute force counted: each block is touched
        assert (want > 0).all()


def test_llama3_launch_shapes(lib):
    """the shapes of the benchmarked model: q|k (Q4_K) + v (Q6_K) merged launch = two runs, wo / down one run of 256 tiles, gate|up 1792"""
    for runs in ([(0, 320), (320, 384)], [(0, 256)], [(0, 1792)], [(0, 8016)]):
        n_blocks = (runs[-1][1] * 16 + 255) // 256
        lo = (ctypes.c_int32 * 3)(*[r[0] for r in runs] + [0] * (3 - len(runs)))
        hi = (ctypes.c_int32 * 3)(*[r[1] for r in runs] + [0] * (3 - len(runs)))
        got = [_f(lib)(len(runs), ctypes.addressof(lo), ctypes.addressof(hi), b) for b in range(n_blocks)]
        assert got == _brute(runs, n_blocks).tolist()
        assert max(got) <= 4 and min(got) >= 1                        # 16 tiles over 7-tile workgroups: 3 or 4 of them (fewer at a run's end)
