"""EXPERIMENTS in probe builds only (tools/build_probe_lib.sh; run with MI355_LIB_PATH=build_probe/libmi355vllm_probes.so
MI355_PROBE_BUILD=1): the single-token mat-vec on the LDS-DMA loader / consumer engine (csrc/probes/qmv_engine.inc) and the four
mat-vecs between two attention calls chained into ONE persistent launch (csrc/probes/qmv_chain.inc; quantized_llama.rs:424-506).
Both measured slower than qmm_kernel launch by launch (profiles/r03_b1_*_probe.txt); what these tests pin is that they compute
the same numbers without a hang: engine == qmm_kernel bit for bit on plain launches, chained greedy tokens == unchained."""
import ctypes
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from oracle import kquants as kq  # noqa: E402


def _probe_or_skip():
    if not os.environ.get("MI355_PROBE_BUILD"):
        pytest.skip("engine / chain are compiled into probe builds only (tools/build_probe_lib.sh)")
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X")


def _err(lib):
    v = ctypes.c_int32(0)
    assert lib.mi355_qmv_error(ctypes.byref(v), 1) == 0
    return v.value


@pytest.mark.parametrize("t", [kq.GGML_Q4_K, kq.GGML_Q6_K])
def test_engine_equals_the_register_ring_kernel(t):
    _probe_or_skip()
    from candle_vllm_amd import ops
    lib = ops.lib
    rng = np.random.default_rng(5)
    for N, K in ((4096, 4096), (272, 14336), (48, 2048)):
        blocks = kq.quantize(rng.normal(0, 0.05, (N, K)).astype(np.float32), t)
        mm = ops.QMatMul(blocks, t, "cuda")
        x = torch.from_numpy(rng.normal(size=(1, K)).astype(np.float32)).cuda()
        try:
            lib.mi355_set_tuning(20, 0)
            ref = mm.forward(x).cpu().numpy()
            for nc in (4, 8):
                lib.mi355_set_tuning(20, 1)
                lib.mi355_set_tuning(21, nc)
                got = mm.forward(x).cpu().numpy()
                assert _err(lib) == 0
                # same unpack + MFMA arithmetic, the partial sums of the k-blocks meet in a different order
                assert np.abs(got - ref).max() <= 1e-5 * np.abs(ref).max()
                assert np.abs(got - kq.qmatmul_o1(x.cpu().numpy(), blocks, t)).max() <= 1e-4 * np.abs(ref).max()
        finally:
            lib.mi355_set_tuning(20, 0)
            lib.mi355_set_tuning(21, 8)


def test_chained_step_equals_launch_by_launch_tokens():
    _probe_or_skip()
    from candle_vllm_amd import model as M
    lib = M.lib
    cfg = M.ModelDims(hidden=2048, n_layers=3, n_heads=16, n_kv_heads=4, head_dim=128, intermediate=4096, vocab=4096)
    bps = 6
    gm = M.GGUFLLaMa(cfg, max_batch=1, max_blocks_per_seq=bps, kv_layout=M.KV_PAGED)
    gm.load_synthetic(seed=11, recipe="q4_k_m")
    gm.alloc_kv_cache(bps + 2)
    stream = torch.cuda.Stream()
    st = stream.cuda_stream
    bt = (np.arange(bps) + 1).reshape(1, bps).astype(np.uint32)

    def run(chain, graph):
        lib.mi355_set_tuning(23, chain)
        gm.set_graph(False)
        gm.set_graph(graph)
        gm.kv_fill_random(seed=7)
        gm.decode_begin(np.array([5], np.uint32), np.full(1, 200, np.uint32), bt, ctx_cap=260, stream=st)
        toks = []
        for _ in range(6):
            gm.decode_step(st)
            toks.append(int(gm.read_tokens(st)[0]))
        return toks, gm.logits_numpy(1)[0].copy()
    try:
        for graph in (False, True):
            t0, l0 = run(0, graph)
            t1, l1 = run(1, graph)
            assert _err(lib) == 0
            assert t0 == t1, (t0, t1)
            assert np.abs(l0 - l1).max() <= 2e-3 * np.abs(l0).max()
    finally:
        lib.mi355_set_tuning(23, 0)


def rel_err(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / (np.abs(np.asarray(b, np.float64)).max() + 1e-30))


@pytest.mark.parametrize("t", [kq.GGML_Q4_K, kq.GGML_Q6_K])
def test_q8k_activation_experiment_equals_the_reference_cpu_numbers(t):
    """mi355_set_tuning(18, 1): single-token launches quantise x to Q8_K and take integer dots on the matrix core, i.e. the
    arithmetic of candle's CPU mat-vec (oracle O2).  Equal to O2 to f32 summation order; O2 itself sits 5e-3..8e-3 away from
    the exact product (O1), which the default path matches to 3e-6."""
    _probe_or_skip()
    from candle_vllm_amd import _lib
    from candle_vllm_amd import ops as cv
    rng = np.random.default_rng(23)
    for N, K in ((64, 1024), (272, 4096), (48, 14336)):
        blocks = kq.quantize(rng.normal(0, 0.05, (N, K)).astype(np.float32), t)
        mm = cv.QMatMul(blocks, t, "cuda")
        x = rng.normal(size=(1, K)).astype(np.float32)
        o1, o2 = kq.qmatmul_o1(x, blocks, t), kq.qmatmul_o2(x, blocks, t)
        try:
            _lib.lib.mi355_set_tuning(18, 1)
            y = mm.forward(torch.from_numpy(x).cuda()).cpu().numpy()
        finally:
            _lib.lib.mi355_set_tuning(18, 0)
        assert rel_err(y, o2) < 1e-6, rel_err(y, o2)
        assert rel_err(mm.forward(torch.from_numpy(x).cuda()).cpu().numpy(), o1) < 1e-5
        assert 1e-3 < rel_err(o2, o1) < 3e-2


