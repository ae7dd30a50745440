"""Error behaviour of the C ABI that is decided on the HOST, before any device call (so it runs without a GPU): empty
inputs are no-ops that return 0 (the reference's ops accept zero-length batches), malformed arguments are refused with
hipErrorInvalidValue (1) instead of launching a kernel that would read out of bounds, and constructors answer NULL.
Calls that would reach the device are not made here -- those are the `-m gpu` tests."""
import ctypes

import pytest

INVALID = 1
BF16, F16, F32 = 2, 1, 0          # MI355_DTYPE_*
KV_FLASH, KV_PAGED = 0, 1


@pytest.fixture(scope="module")
def L(lib):
    from candle_vllm_amd import _lib
    assert (_lib.lib.mi355_abi_struct_size(0)) > 0
    return _lib


def test_dtype_codes_match_the_header():
    import os
    import re
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "mi355_vllm.h")).read()
    codes = {k: int(v) for k, v in re.findall(r"#define (MI355_DTYPE_\w+|MI355_KV_\w+) (\d+)", hdr)}
    assert (codes["MI355_DTYPE_F32"], codes["MI355_DTYPE_F16"], codes["MI355_DTYPE_BF16"]) == (F32, F16, BF16)
    assert (codes["MI355_KV_FLASH"], codes["MI355_KV_PAGED"]) == (KV_FLASH, KV_PAGED)


def test_empty_inputs_are_no_ops(L):
    lib = L.lib
    assert lib.mi355_reshape_and_cache(None, None, None, None, None, 0, 8, 128, 64, 2, KV_PAGED, 0) == 0
    assert lib.mi355_reshape_and_cache_fp8(None, None, None, None, None, 0, 8, 128, 64, KV_PAGED, 1.0, 1.0, 0) == 0
    assert lib.mi355_swap_blocks(None, None, None, 0, 131072, 0, 0) == 0
    assert lib.mi355_swap_blocks(None, None, None, 4, 0, 0, 0) == 0
    # zero sequences: nothing to attend
    assert lib.mi355_paged_attention_v1(None, None, None, None, None, None, 0, 32, 8, 128, 64, 8, 512, 0.088, 0.0,
                                        KV_PAGED, BF16, 0) == 0


def test_malformed_attention_arguments_are_refused(L):
    lib = L.lib
    one = ctypes.c_uint64(0)
    p = ctypes.addressof(one)                                  # any non-null pointer: the call must fail before using it

    def v1(H=32, Hkv=8, D=128, layout=KV_PAGED, dtype=BF16):
        return lib.mi355_paged_attention_v1(p, p, p, p, p, p, 1, H, Hkv, D, 64, 8, 512, 0.088, 0.0, layout, dtype, 0)
    assert v1(H=30) == INVALID                                 # query heads not a multiple of kv heads
    assert v1(D=100) == INVALID and v1(D=512) == INVALID       # head size: multiple of 8, <= 256
    assert v1(dtype=F32) == INVALID and v1(dtype=7) == INVALID

    def v2(ps, tmp=p):
        return lib.mi355_paged_attention_v2(p, tmp, tmp, tmp, p, p, p, p, p, 1, 32, 8, 128, 64, 8, 512, ps, 0.088, 0.0,
                                            KV_PAGED, BF16, 0)
    assert v2(0) == INVALID and v2(-32) == INVALID             # v2 needs a partition size
    assert lib.mi355_paged_attention_v2(p, None, None, None, p, p, p, p, p, 1, 30, 8, 128, 64, 8, 512, 32, 0.088, 0.0,
                                        KV_PAGED, BF16, 0) == INVALID
    # fp8 cache: partitioned call without the partial buffers
    assert lib.mi355_paged_attention_fp8(p, None, None, None, p, p, p, p, p, 1, 32, 8, 128, 64, 8, 512, 32, 0.088, 0.0,
                                         1.0, 1.0, 0) == INVALID
    # prefill: dtype, head grouping, missing k/v when there is no cache, missing tables when there is one
    def pre(k=p, v=p, kc=None, vc=None, bt=None, cl=None, H=32, Hkv=8, D=128, dtype=BF16):
        return lib.mi355_prefill_attention(p, p, k, v, kc, vc, bt, cl, p, 1, 16, H, Hkv, D, 64, 8, 0.088, 0.0, KV_PAGED, dtype, 0)
    assert pre(dtype=F32) == INVALID and pre(H=30) == INVALID and pre(D=320) == INVALID
    assert pre(k=None, v=None) == INVALID
    assert pre(kc=p, vc=p, bt=None, cl=None) == INVALID


def test_malformed_cache_and_elementwise_arguments_are_refused(L):
    lib = L.lib
    one = ctypes.c_uint64(0)
    p = ctypes.addressof(one)
    assert lib.mi355_reshape_and_cache(p, p, p, p, p, 1, 8, 128, 64, 3, KV_PAGED, 0) == INVALID      # element size
    assert lib.mi355_reshape_and_cache(p, p, p, p, p, 1, 8, 100, 64, 2, KV_PAGED, 0) == INVALID      # D % x
    assert lib.mi355_reshape_and_cache(p, p, p, p, p, 1, 8, 128, 64, 2, 9, 0) == INVALID             # layout code
    assert lib.mi355_reshape_and_cache_fp8(p, p, p, p, p, 1, 8, 120, 64, KV_PAGED, 1.0, 1.0, 0) == INVALID
    assert lib.mi355_swap_blocks(p, p, p, 1, 4096, 99, 0) == INVALID                                 # direction code
    assert lib.mi355_rope_inplace(p, p, p, p, p, 1, 4, 2, 128, 0, 0, BF16, 0) == INVALID             # rotary_dim 0
    assert lib.mi355_rope_inplace(p, p, p, p, p, 1, 4, 2, 128, 130, 0, BF16, 0) == INVALID           # > head_dim
    assert lib.mi355_rope_inplace(p, p, p, p, p, 1, 4, 2, 128, 63, 0, BF16, 0) == INVALID            # odd
    assert lib.mi355_argmax_f32(p, p, 1, 0, 0) == INVALID
    assert lib.mi355_moe_route(p, p, p, p, 1e-5, p, 1, 4096, 8, 9, 0) == INVALID                     # top-k > experts
    assert lib.mi355_moe_route(p, p, p, p, 1e-5, p, 1, 4096, 0, 1, 0) == INVALID


def test_malformed_quantised_matmul_descriptors_are_refused(L):
    lib = L.lib
    one = ctypes.c_uint64(0)
    p = ctypes.addressof(one)
    assert lib.mi355_qmatmul_fused(None, 0) == INVALID
    d = L.QmmDesc()
    d.nseg = 0
    assert lib.mi355_qmatmul_fused(ctypes.byref(d), 0) == INVALID
    d.nseg = 4
    assert lib.mi355_qmatmul_fused(ctypes.byref(d), 0) == INVALID
    d = L.QmmDesc()
    d.nseg, d.x_dtype = 1, F16                                  # activations are f32 or bf16
    assert lib.mi355_qmatmul_fused(ctypes.byref(d), 0) == INVALID
    d = L.QmmDesc()
    d.nseg, d.x_dtype, d.epilogue = 1, F32, 1                   # MI355_EPI_RESID without a residual pointer
    d.out = p
    assert lib.mi355_qmatmul_fused(ctypes.byref(d), 0) == INVALID
    d.epilogue, d.out = 0, None                                 # MI355_EPI_STORE without an output pointer
    assert lib.mi355_qmatmul_fused(ctypes.byref(d), 0) == INVALID
    # grouped calls (group_count mat-muls of the same shapes): a negative count or stride, or grouping together with per-pair expert ids
    d = L.QmmDesc()
    d.nseg, d.x_dtype, d.epilogue, d.out, d.group_count = 1, F32, 0, p, -1
    assert lib.mi355_qmatmul_fused(ctypes.byref(d), 0) == INVALID
    d.group_count, d.group_x_stride = 4, -1
    assert lib.mi355_qmatmul_fused(ctypes.byref(d), 0) == INVALID
    d.group_x_stride, d.moe_expert_ids = 0, p
    assert lib.mi355_qmatmul_fused(ctypes.byref(d), 0) == INVALID
    # the fused MoE staging / scatter launches of the host layer say "not this configuration" (-4) before anything touches the device
    lib.mi355_internal_moe_stage_grouped.restype = ctypes.c_int
    lib.mi355_internal_moe_stage_grouped.argtypes = [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int32] * 7 + [ctypes.c_void_p] * 4 + [ctypes.c_int32, ctypes.c_int64]
    assert lib.mi355_internal_moe_stage_grouped(None, None, 64, 2, 8, 64, 0, 32, 4096, None, None, None, None, 32, 0) == -4
    assert lib.mi355_internal_moe_stage_grouped(p, p, 64, 2, 8, 32, 0, 32, 4096, None, p, p, p, 32, 0) == -4          # cap < pairs
    assert lib.mi355_internal_moe_stage_grouped(p, p, 64, 2, 8, 64, 0, 5, 4096, None, p, p, p, 32, 0) == -4           # 5 rows: not the 9..32-token path
    lib.mi355_internal_moe_scatter_combine_to_image.restype = ctypes.c_int
    lib.mi355_internal_moe_scatter_combine_to_image.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int32] * 3 + [ctypes.c_void_p, ctypes.c_int64]
    assert lib.mi355_internal_moe_scatter_combine_to_image(p, p, p, p, 4, 4096, 2, None, 0) == -4                 # 4 tokens
    assert lib.mi355_internal_moe_scatter_combine_to_image(p, p, p, p, 32, 4000, 2, None, 0) == -4                # hidden % 256
    # repack: sizes are host arithmetic
    assert lib.mi355_qweight_repacked_size(12, 16, 256) == 2304 and lib.mi355_qweight_repacked_size(14, 16, 256) == 3360
    assert lib.mi355_qweight_repacked_size(12, 16, 100) < 0 and lib.mi355_qweight_repacked_size(8, 16, 256) < 0
    assert lib.mi355_qweight_repacked_size(12, 17, 256) == 2 * 2304        # rows are padded to whole 16-row tiles
    assert lib.mi355_qweight_repack(p, p, 8, 16, 256) != 0


def test_constructors_refuse_bad_configurations(L):
    lib = L.lib
    assert not lib.mi355_llama_create(None)
    c = L.LlamaConfig()
    assert not lib.mi355_llama_create(ctypes.byref(c))          # all zero
    c.hidden, c.n_layers, c.n_heads, c.n_kv_heads, c.head_dim, c.max_batch, c.max_blocks_per_seq = 250, 1, 2, 2, 64, 1, 4
    assert not lib.mi355_llama_create(ctypes.byref(c))          # hidden not a multiple of the 256-wide k-block
    assert not lib.mi355_dense_create(None)
    lib.mi355_llama_destroy(None)
    lib.mi355_dense_destroy(None)
    # communicator entry points on a null handle
    one = ctypes.c_uint64(0)
    assert lib.mi355_comm_all_reduce(None, ctypes.addressof(one), 1, F32, 0) != 0
    assert lib.mi355_comm_all_gather(None, ctypes.addressof(one), ctypes.addressof(one), 1, F32, 0) != 0
    lib.mi355_comm_destroy(None)
    assert lib.mi355_llama_set_comm(None, None) != 0


def test_scoped_tuning_restores_what_it_found(lib):
    """candle_vllm_amd.tuning(key, value): the key holds the value inside the block and what it held before afterwards, also when
    the block raises (the A/B switches are process-global state of the library)"""
    from candle_vllm_amd import tuning
    from candle_vllm_amd import TuningKeyUnavailable
    product = [3, 5, 6, 9, 24, 30, 41, 44, 47, 48]                # the ten keys of the product library (include/mi355_vllm.h)
    defaults = {3: 1, 5: 0, 6: 1, 9: 1, 24: 0, 30: 0, 41: 1, 44: 1, 47: 1, 48: 1}
    for k in range(64):
        assert bool(lib.mi355_tuning_supported(k)) == (k in product), k      # (a probe build honours all 64: this suite runs the product)
        if k in product:
            assert lib.mi355_get_tuning(k) == defaults[k], k     # the library reports the live value of every product key
    assert lib.mi355_get_tuning(63) == -2 ** 31                   # a probe-build key: never set ...
    lib.mi355_set_tuning(63, 5)
    assert lib.mi355_get_tuning(63) == -2 ** 31                   # ... and ignored by the product library
    with pytest.raises(TuningKeyUnavailable):
        with tuning(63, 5):
            pass
    with tuning(5, 64):
        assert lib.mi355_get_tuning(5) == 64
        with tuning(5, 128):
            assert lib.mi355_get_tuning(5) == 128
        assert lib.mi355_get_tuning(5) == 64
    assert lib.mi355_get_tuning(5) == 0
    try:
        with tuning(41, 0):
            assert lib.mi355_get_tuning(41) == 0
            raise ValueError
    except ValueError:
        pass
    assert lib.mi355_get_tuning(41) == 1                          # restored to its (non-zero) default
    assert lib.mi355_get_tuning(-1) == -2 ** 31 and lib.mi355_get_tuning(64) == -2 ** 31
