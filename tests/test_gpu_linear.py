"""GPU parity tests for the safetensors path (K10, K12-K15): dense 16-bit and GPTQ/AWQ/Marlin 4-bit linears through
the C ABI vs the numpy oracle.  Repack / permutation results are bit-exact; matmul outputs are 16-bit values
compared at 1 ulp of the output dtype (tolerance next to the assert)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from oracle import gptq as G               # noqa: E402

TD = {"bf16": torch.bfloat16, "f16": torch.float16}
ULP = {"bf16": 2.0 ** -8, "f16": 2.0 ** -11}


@pytest.fixture(scope="module")
def cv(lib):
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X (torch.cuda.is_available() is False)")
    import candle_vllm_amd.ops as ops
    return ops


def dev16(a_f32, dt):
    """f32 numpy values (already exactly representable) -> 16-bit cuda tensor, bit-exact"""
    bits = G.to_bits(a_f32, dt)
    return torch.from_numpy(np.ascontiguousarray(bits).view(np.int16)).cuda().view(TD[dt])


def host16(t, dt):
    return G.from_bits(t.detach().view(torch.int16).cpu().numpy().view(np.uint16), dt)


def dev_u32(a):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).cuda()


def check_ulp(got, ref, dt, ulps=1.01, what="", mag=None):
    """|got-ref| <= ulps * ulp(mag) elementwise (mag = |ref|, or the largest intermediate of a rounding chain:
    a 1-ulp flip of the matmul result survives a cancelling bias / residual add), plus a small absolute floor for
    cancellation inside the dot product (f32 accumulation order differs from the f64 oracle)."""
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    mag = np.abs(ref) if mag is None else np.maximum(np.abs(ref), np.abs(mag))
    tol = ulps * 2 * ULP[dt] * mag                              # ulp(v) <= 2 * 2^-p * |v|
    floor = 1e-3 * np.abs(ref).max() * (ULP[dt] / ULP["bf16"])
    bad = np.abs(got - ref) > np.maximum(tol, floor)
    assert not bad.any(), f"{what}: {bad.sum()} / {bad.size} off, max abs err {np.abs(got - ref).max()}"


# ------------------------------------------------------------------------------------------------ K10 dense
@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("T,N,K", [(1, 256, 512), (5, 48, 256), (16, 1024, 1024), (17, 64, 768), (33, 128, 512),
                                   (64, 96, 256), (70, 32, 512), (128, 64, 512), (200, 48, 256),
                                   (256, 384, 1024), (300, 200, 576), (97, 130, 64)])   # >= 96 tokens: the hand-written 128 x 128 MFMA GEMM (ragged T / N tiles, one K step)
def test_linear_matches_oracle(cv, dt, T, N, K):
    rng = np.random.default_rng(T * 1000 + N)
    x = G.round_dt(rng.normal(0, 1, (T, K)), dt)
    w = G.round_dt(rng.normal(0, 0.05, (N, K)), dt)
    b = G.round_dt(rng.normal(0, 0.2, N), dt)
    lin = cv.Linear(dev16(w, dt), dev16(b, dt))
    y = host16(lin.forward(dev16(x, dt)), dt)
    check_ulp(y, G.linear16(x, w, b, dt), dt, what="linear+bias", ulps=2.01, mag=G.linear16(x, w, None, dt))
    lin2 = cv.Linear(dev16(w, dt))
    res = G.round_dt(rng.normal(0, 1, (T, N)), dt)
    y2 = host16(lin2.forward(dev16(x, dt), epilogue=cv.EPI_RESID, residual=dev16(res, dt)), dt)
    check_ulp(y2, G.round_dt(G.linear16(x, w, None, dt) + res, dt), dt, what="linear+resid", ulps=2.01,
              mag=G.linear16(x, w, None, dt))


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("T,N,K", [(256, 384, 1024), (300, 200, 576), (97, 130, 64), (700, 128, 256)])
def test_linear_prompt_gemm_tall_tile(cv, dt, T, N, K):
    """the 256 x 128 tile of the 16-bit prompt GEMM (8 waves, tuning key 30 bits 32 | 64 = wherever the GEMM runs): the same rounding chain as
    the 128 x 128 tile -- bit for bit (same K order per output element) -- and within the op's bound of the oracle; plain, tiled-image and
    SiLU * up launches, ragged token / column tiles"""
    from candle_vllm_amd import tuning
    rng = np.random.default_rng(T * 7 + N)
    x = G.round_dt(rng.normal(0, 1, (T, K)), dt)
    w = G.round_dt(rng.normal(0, 0.05, (N, K)), dt)
    b = G.round_dt(rng.normal(0, 0.2, N), dt)
    res = G.round_dt(rng.normal(0, 1, (T, N)), dt)
    I = N // 2
    lin, lin2 = cv.Linear(dev16(w, dt), dev16(b, dt)), cv.Linear(dev16(w, dt))

    def run():
        y = host16(lin.forward(dev16(x, dt)), dt)
        y2 = host16(lin2.forward(dev16(x, dt), epilogue=cv.EPI_RESID, residual=dev16(res, dt)), dt)
        y3 = host16(lin2.forward(dev16(x, dt), epilogue=cv.EPI_SILU_MUL), dt)
        return y, y2, y3

    base = run()
    with tuning(30, 32 | 64):
        tall = run()
    for a_, b_ in zip(base, tall):
        assert np.array_equal(np.asarray(a_), np.asarray(b_))
    check_ulp(tall[0], G.linear16(x, w, b, dt), dt, what="linear+bias", ulps=2.01, mag=G.linear16(x, w, None, dt))
    gu = G.linear16(x, w, None, dt)
    check_ulp(tall[2], G.silu_mul16(gu[:, :I], gu[:, I:], dt), dt, ulps=3.0, what="silu(gate)*up")


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("T", [1, 8, 32, 40, 130, 257])
def test_linear_gate_up_silu(cv, dt, T):
    rng = np.random.default_rng(T)
    K, I = (512, 192) if T != 257 else (1024, 328)              # 257 tokens: three token tiles, 5.1 column tiles of the prompt GEMM
    x = G.round_dt(rng.normal(0, 1, (T, K)), dt)
    w = G.round_dt(rng.normal(0, 0.06, (2 * I, K)), dt)          # packed [gate; up] (mlp.rs:324-352)
    lin = cv.Linear(dev16(w, dt))
    y = host16(lin.forward(dev16(x, dt), epilogue=cv.EPI_SILU_MUL), dt)
    gu = G.linear16(x, w, None, dt)
    ref = G.silu_mul16(gu[:, :I], gu[:, I:], dt)
    # the 1-ulp differences of gate/up (f32 vs f64 accumulation) propagate through silu*up: 3 ulp
    check_ulp(y, ref, dt, ulps=3.0, what="silu(gate)*up")


# ------------------------------------------------------------------------------------------------ K15 repack
def test_repack_bit_exact(cv):
    rng = np.random.default_rng(3)
    K, N = 512, 128
    q = rng.integers(0, 16, (K, N))
    gq = G.gptq_pack(q)
    tiled = G.gptq_tile(gq)                                       # numpy statement of the library's 16 x 256 tile order
    out = cv.marlin_weight_repack(dev_u32(gq), 4, False)
    assert tuple(out.shape) == (K // 16, 2 * N)                   # gptq.rs:291-297
    assert (out.cpu().numpy().view(np.uint32).reshape(-1) == tiled).all()
    assert (cv.gptq_tile_unpack(out, K, N).cpu().numpy().view(np.uint32) == gq).all()      # and back: a permutation of words
    out2 = cv.marlin_weight_repack(dev_u32(G.awq_pack(q)), 4, True)
    assert tuple(out2.shape) == (K // 16, 2 * N)                  # gptq.rs:285-290
    assert (out2.cpu().numpy().view(np.uint32).reshape(-1) == tiled).all()
    # a shape the marlin arm does not take (k % 256 != 0) keeps the checkpoint layout
    q3 = rng.integers(0, 16, (128, 32))
    out3 = cv.marlin_weight_repack(dev_u32(G.gptq_pack(q3)), 4, False)
    assert (out3.cpu().numpy().view(np.uint32).reshape(16, 32) == G.gptq_pack(q3)).all()
    # columns [0, n) into tiles tile0 .. of a wider image (how the host layer packs gate_proj | up_proj)
    from candle_vllm_amd import lib
    wide = torch.zeros((K // 8) * 2 * N, dtype=torch.int32, device="cuda")
    q4 = rng.integers(0, 16, (K, N))
    parts = [dev_u32(gq), dev_u32(G.gptq_pack(q4))]
    for part, t0 in zip(parts, (0, N // 16)):
        assert lib.mi355_gptq_tile_repack(part.data_ptr(), wide.data_ptr(), K, N, t0, 0) == 0
    torch.cuda.synchronize()
    both = np.concatenate([gq, G.gptq_pack(q4)], axis=1)
    assert (wide.cpu().numpy().view(np.uint32) == G.gptq_tile(both)).all()


# ------------------------------------------------------------------------------------------------ K12 / K13 marlin
def _gptq_case(rng, K, N, gs, dt):
    q = rng.integers(0, 16, (K, N))
    ng = 1 if gs == -1 else K // gs
    s = G.round_dt(rng.uniform(0.005, 0.02, (ng, N)), dt)
    return q, s


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("T,N,K,gs", [(1, 128, 512, 128), (3, 64, 256, 64), (16, 256, 1024, 128), (20, 64, 512, -1),
                                      (33, 128, 512, 32), (64, 64, 768, 256), (7, 192, 512, 64)])
def test_marlin_4bit_matches_oracle(cv, dt, T, N, K, gs):
    rng = np.random.default_rng(K + N + T)
    q, s = _gptq_case(rng, K, N, gs, dt)
    x = G.round_dt(rng.normal(0, 1, (T, K)), dt)
    qw = cv.marlin_weight_repack(dev_u32(G.gptq_pack(q)), 4, False)
    sp = G.marlin_permute_scales(s, K, N, gs)                    # what the reference's loader hands over
    ws = torch.zeros(N, dtype=torch.int32, device="cuda")
    y = cv.gptq_matmul(dev16(x, dt).reshape(1, T, K), qw, dev16(sp, dt), None, None, ws, 4, gs, False)
    assert tuple(y.shape) == (1, T, N)
    ref = G.gptq_linear(x, G.gptq_dequant(q, s, None, gs), None, dt)
    check_ulp(host16(y, dt).reshape(T, N), ref, dt, what="marlin_4bit")


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("T,N,K,gs", [(1, 128, 512, 128), (9, 64, 256, 64), (32, 192, 512, 128)])
def test_marlin_awq_matches_oracle(cv, dt, T, N, K, gs):
    rng = np.random.default_rng(K + N + T + 1)
    q, s = _gptq_case(rng, K, N, gs, dt)
    z = rng.integers(0, 16, (K // gs, N))
    x = G.round_dt(rng.normal(0, 1, (T, K)), dt)
    qw = cv.marlin_weight_repack(dev_u32(G.awq_pack(q)), 4, True)
    awq_qzeros = G.awq_pack(z)                                   # AWQ checkpoint packing
    mzp = G.awq_to_marlin_zero_points(awq_qzeros, K // gs, N)    # examples/convert_awq_marlin.py
    sp = G.marlin_permute_scales(s, K, N, gs)
    ws = torch.zeros(N, dtype=torch.int32, device="cuda")
    y = cv.gptq_matmul(dev16(x, dt).reshape(1, T, K), qw, dev16(sp, dt), dev_u32(mzp), None, ws, 4, gs, True)
    ref = G.gptq_linear(x, G.gptq_dequant(q, s, z, gs), None, dt)
    check_ulp(host16(y, dt).reshape(T, N), ref, dt, what="marlin_awq_4bit")


# ------------------------------------------------------------------------------------------------ K14 exllama
@pytest.mark.parametrize("T,N,K,gs", [(1, 64, 256, 64), (5, 128, 512, 128), (24, 64, 256, 32), (40, 32, 128, 32)])
def test_exllama_act_order_matches_oracle(cv, T, N, K, gs):
    rng = np.random.default_rng(K + T)
    q, s = _gptq_case(rng, K, N, gs, "f16")
    z = rng.integers(1, 17, (K // gs, N))
    g_idx = (rng.permutation(K) // gs).astype(np.int32)          # desc_act
    x = G.round_dt(rng.normal(0, 1, (T, K)), "f16")
    y = cv.gptq_matmul(dev16(x, "f16").reshape(1, T, K), dev_u32(G.gptq_pack(q)), dev16(s, "f16"),
                       dev_u32(G.gptq_pack_zeros(z)), torch.from_numpy(g_idx).cuda(), None, 4, gs, False)
    w16 = G.gptq_dequant(q, s, z, g_idx=g_idx, round_to="f16")   # half-precision dequant, as the exllama family does
    ref = G.gptq_linear(x, w16, None, "f16")
    check_ulp(host16(y, "f16").reshape(T, N), ref, "f16", what="gemm_half_q_half_alt")
    # and within north_star's 1e-3 of the exact (unrounded-weight) definition
    exact = x.astype(np.float64) @ G.gptq_dequant(q, s, z, g_idx=g_idx)
    assert np.abs(host16(y, "f16").reshape(T, N) - exact).max() <= 2e-3 * np.abs(exact).max()


@pytest.mark.parametrize("T,N,K,gs", [(1, 64, 256, 64), (7, 128, 512, 128), (33, 32, 128, 32)])
def test_exllama_8bit_matches_oracle(cv, T, N, K, gs):
    """bits = 8 on the exllama arm (the reference admits 4 | 8, linear.rs:215-217, and forwards `bits`, gptq.rs:181-194)"""
    rng = np.random.default_rng(K + T + 8)
    q = rng.integers(0, 256, (K, N))
    s = G.round_dt(rng.uniform(0.0005, 0.002, (K // gs, N)), "f16")
    z = rng.integers(1, 257, (K // gs, N))
    g_idx = (rng.permutation(K) // gs).astype(np.int32)
    x = G.round_dt(rng.normal(0, 1, (T, K)), "f16")
    y = cv.gptq_matmul(dev16(x, "f16").reshape(1, T, K), dev_u32(G.gptq_pack8(q)), dev16(s, "f16"),
                       dev_u32(G.gptq_pack_zeros8(z)), torch.from_numpy(g_idx).cuda(), None, 8, gs, False)
    assert tuple(y.shape) == (1, T, N)
    w16 = G.gptq_dequant(q, s, z, g_idx=g_idx, round_to="f16")
    ref = G.gptq_linear(x, w16, None, "f16")
    check_ulp(host16(y, "f16").reshape(T, N), ref, "f16", what="gemm_half_q_half_alt 8-bit")


def test_void_ffi_refuses_loudly(cv):
    """a configuration the library cannot serve must not read as data: NaN-filled output + a recorded error
    (`bits != 4` used to return with `c` untouched)"""
    from candle_vllm_amd import lib
    rng = np.random.default_rng(5)
    T, N, K, gs = 3, 64, 256, 64
    q, s = _gptq_case(rng, K, N, gs, "f16")
    z = rng.integers(1, 17, (K // gs, N))
    x = dev16(G.round_dt(rng.normal(0, 1, (T, K)), "f16"), "f16")
    out = torch.zeros((T, N), dtype=torch.float16, device="cuda")
    lib.mi355_clear_error()
    lib.gemm_half_q_half_alt(x.data_ptr(), dev_u32(G.gptq_pack(q)).data_ptr(), dev_u32(G.gptq_pack_zeros(z)).data_ptr(),
                             dev16(s, "f16").data_ptr(), None, out.data_ptr(), T, N, K, 3, 0)      # 3-bit: not served
    torch.cuda.synchronize()
    assert lib.mi355_last_error() != 0
    assert torch.isnan(out).all()
    lib.mi355_clear_error()
    assert lib.mi355_last_error() == 0
    with pytest.raises(RuntimeError):                             # the Python mirror turns the record into an exception
        cv.gptq_matmul(x.reshape(1, T, K), dev_u32(G.gptq_pack(q)), dev16(s, "f16"), dev_u32(G.gptq_pack_zeros(z)),
                       torch.zeros(K, dtype=torch.int32, device="cuda"), None, 3, gs, False)
    assert lib.mi355_last_error() == 0


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("T,N,K,gs", [(1, 128, 512, 128), (16, 256, 1024, 128), (5, 64, 256, -1)])
def test_marlin_format_checkpoint(cv, dt, T, N, K, gs):
    """`checkpoint_format == "marlin"`: B [k/16, 2n] straight from the file (Marlin tile order) + permuted `s`; the load-time
    un-permute makes it the weight image marlin_4bit_* streams, bit for bit the image gptq_repack makes of the same codes."""
    rng = np.random.default_rng(K + N + T + 77)
    q, s = _gptq_case(rng, K, N, gs, dt)
    B = G.marlin_format_pack(q)
    assert B.shape == (K // 16, 2 * N)                            # linear.rs:226-231 (marlin_format dims)
    qw = cv.marlin_format_repack(dev_u32(B))
    if K % 256 == 0:
        assert (qw.cpu().numpy().view(np.uint32).reshape(-1) == G.gptq_tile(G.gptq_pack(q))).all()
    else:
        assert (qw.cpu().numpy().view(np.uint32).reshape(K // 8, N) == G.gptq_pack(q)).all()
    x = G.round_dt(rng.normal(0, 1, (T, K)), dt)
    sp = G.marlin_permute_scales(s, K, N, gs)
    ws = torch.zeros(N, dtype=torch.int32, device="cuda")
    y = cv.gptq_matmul(dev16(x, dt).reshape(1, T, K), qw, dev16(sp, dt), None, None, ws, 4, gs, False)
    ref = G.gptq_linear(x, G.gptq_dequant(q, s, None, gs), None, dt)
    check_ulp(host16(y, dt).reshape(T, N), ref, dt, what="marlin-format checkpoint")


# ------------------------------------------------------------------------------------------------ fused GPTQ epilogues
@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("T", [6, 1, 4])
def test_gptq_linear_fused_epilogues(cv, dt, T):
    """T = 6: the 16-token-tile kernel; T = 1, 4: the 1..4-token kernel (activations staged in LDS, weight ring)"""
    rng = np.random.default_rng(11)
    K, I, gs = 512, 128, 128
    q, s = _gptq_case(rng, K, 2 * I, gs, dt)
    x = G.round_dt(rng.normal(0, 1, (T, K)), dt)
    lin = cv.GPTQLinear(dev_u32(G.gptq_pack(q)), dev16(s, dt), gs)
    y = host16(lin.forward(dev16(x, dt), epilogue=cv.EPI_SILU_MUL), dt)
    gu = G.gptq_linear(x, G.gptq_dequant(q, s, None, gs), None, dt)
    check_ulp(y, G.silu_mul16(gu[:, :I], gu[:, I:], dt), dt, ulps=3.0, what="gptq silu*up")
    z = rng.integers(1, 17, (K // gs, 2 * I))
    lin2 = cv.GPTQLinear(dev_u32(G.gptq_pack(q)), dev16(s, dt), gs, qzeros=dev_u32(G.gptq_pack_zeros(z)),
                         zero_mode=cv.ZERO_GPTQ_PLUS1)
    res = G.round_dt(rng.normal(0, 1, (T, 2 * I)), dt)
    y2 = host16(lin2.forward(dev16(x, dt), epilogue=cv.EPI_RESID, residual=dev16(res, dt)), dt)
    ref2 = G.round_dt(G.gptq_linear(x, G.gptq_dequant(q, s, z, gs), None, dt) + res, dt)
    check_ulp(y2, ref2, dt, what="gptq asym + resid", ulps=2.01, mag=G.gptq_linear(x, G.gptq_dequant(q, s, z, gs), None, dt))


@pytest.mark.parametrize("tiled", [True, False])
@pytest.mark.parametrize("T,N,K,gs", [(1, 256, 3584, 128), (2, 64, 3584, 64), (3, 64, 1792, 32), (1, 48, 18944, 128), (4, 32, 4864, 256)])
def test_small_batch_4bit_kernel_shapes(cv, T, N, K, gs, tiled):
    """the 1..4-token kernel at k-block counts that do not divide by 8 (Qwen2: 14 and 74), every group size, both weight
    layouts, against the oracle -- and against the 16-token-tile kernel (tuning key 31) to the same bound"""
    from candle_vllm_amd import lib
    rng = np.random.default_rng(K + N + T)
    q, s = _gptq_case(rng, K, N, gs, "bf16")
    x = G.round_dt(rng.normal(0, 1, (T, K)), "bf16")
    lin = cv.GPTQLinear(dev_u32(G.gptq_pack(q)), dev16(s, "bf16"), gs, tiled=tiled)
    ref = G.gptq_linear(x, G.gptq_dequant(q, s, None, gs), None, "bf16")
    y = host16(lin.forward(dev16(x, "bf16")), "bf16")
    check_ulp(y, ref, "bf16", what="1..4-token kernel")
    from candle_vllm_amd import tuning
    with tuning(30, 1):                                           # key 30 bit 0: the 1..4-token kernel off
        y0 = host16(lin.forward(dev16(x, "bf16")), "bf16")
    check_ulp(y0, ref, "bf16", what="16-token-tile kernel")


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("T,N,K,pair", [(32, 12304, 512, False), (9, 12288, 256, False), (64, 12320, 768, False), (48, 12288, 512, False),
                                        (32, 12304, 512, True), (5, 12288, 256, True), (20, 12320, 768, True)])
def test_wide_16bit_kernel_many_row_tiles(cv, dt, T, N, K, pair):
    """5..64 tokens against >= 768 row tiles (gate/up, lm_head shapes): the LDS-shared-activation kernel (one row tile or gate/up
    pair per wave, 2-slot weight ring; a tile count that is not a multiple of 4 leaves surplus waves), against the oracle -- and
    the K-split kernel on the same inputs (tuning key 37) to the same bound"""
    from candle_vllm_amd import lib
    rng = np.random.default_rng(T * 31 + N + K)
    x = G.round_dt(rng.normal(0, 1, (T, K)), dt)
    if pair:
        w = G.round_dt(rng.normal(0, 0.06, (2 * N, K)), dt)
        gu = G.linear16(x, w, None, dt)
        ref, mag, ulps = G.silu_mul16(gu[:, :N], gu[:, N:], dt), None, 3.0
    else:
        w = G.round_dt(rng.normal(0, 0.05, (N, K)), dt)
        ref, mag, ulps = G.linear16(x, w, None, dt), None, 1.01
    lin = cv.Linear(dev16(w, dt))
    kw = {"epilogue": cv.EPI_SILU_MUL} if pair else {}
    y = host16(lin.forward(dev16(x, dt), **kw), dt)
    check_ulp(y, ref, dt, ulps=ulps, what="wide 16-bit kernel", mag=mag)
    from candle_vllm_amd import tuning
    with tuning(30, 8):                                           # key 30 bit 3: the LDS-shared-activation kernel off
        y0 = host16(lin.forward(dev16(x, dt), **kw), dt)
    check_ulp(y0, ref, dt, ulps=ulps, what="K-split kernel", mag=mag)


def test_dense_tile_repack_bit_exact(cv):
    from candle_vllm_amd import lib
    rng = np.random.default_rng(8)
    N, K = 64, 768
    w = rng.integers(0, 2 ** 16, (N, K), dtype=np.uint16)
    d_in = torch.from_numpy(w.view(np.int16)).cuda()
    d_out = torch.zeros_like(d_in)
    assert lib.mi355_dense_tile_repack(d_in.data_ptr(), d_out.data_ptr(), N, K, K, 0) == 0
    torch.cuda.synchronize()
    assert (d_out.cpu().numpy().view(np.uint16).reshape(-1) == G.dense_tile(w)).all()
    assert lib.mi355_dense_tile_repack(d_in.data_ptr(), d_out.data_ptr(), 40, K, K, 0) != 0      # n % 16
    assert lib.mi355_dense_tile_repack(d_in.data_ptr(), d_out.data_ptr(), N, 128, 128, 0) != 0    # k % 256


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("T,N,K,pair", [(1, 64, 512, False), (8, 128, 256, False), (33, 96, 768, False), (64, 32, 512, False),
                                        (32, 12304, 512, False), (20, 12320, 256, True), (3, 96, 512, True),
                                        (130, 208, 512, False), (257, 336, 1024, True), (97, 144, 256, False)])
def test_linear_over_the_tiled_weight_image(cv, dt, T, N, K, pair):
    """mi355_linear_tiled through all three 16-bit kernels (K-split streaming kernel, LDS-shared-activation kernel at >= 768 row tiles,
    128 x 128 MFMA GEMM from 96 tokens) == the oracle == the row-major result to the same bound"""
    rng = np.random.default_rng(T * 7 + N + K)
    x = G.round_dt(rng.normal(0, 1, (T, K)), dt)
    if pair:
        w = G.round_dt(rng.normal(0, 0.06, (2 * N, K)), dt)
        gu = G.linear16(x, w, None, dt)
        ref, ulps, kw = G.silu_mul16(gu[:, :N], gu[:, N:], dt), 3.0, {"epilogue": cv.EPI_SILU_MUL}
    else:
        w = G.round_dt(rng.normal(0, 0.05, (N, K)), dt)
        ref, ulps, kw = G.linear16(x, w, None, dt), 1.01, {}
    for tiled in (True, False):
        lin = cv.Linear(dev16(w, dt), tiled=tiled)
        y = host16(lin.forward(dev16(x, dt), **kw), dt)
        check_ulp(y, ref, dt, ulps=ulps, what=f"linear tiled={tiled}")


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("T,N,K,gs,mode", [(96, 64, 256, 64, "sym"), (130, 208, 512, 128, "sym"), (257, 96, 1024, 256, "sym"),
                                           (200, 48, 512, -1, "sym"), (128, 80, 768, 128, "zp"), (161, 144, 512, 128, "silu"),
                                           (97, 32, 256, 128, "resid")])
def test_gptq_prompt_gemm_one_pass(cv, dt, T, N, K, gs, mode):
    """>= 96 tokens over the tiled 4-bit image: the one-pass MFMA GEMM (weights unpacked in registers, per-group scale / offset with
    the prep launch's group sums of x; ragged token and column tiles, groups of 64 / 128 / 256 / whole K, zero points, the fused
    epilogues) == the oracle == the decode kernel in 64-token chunks (tuning key 39) to the same bound"""
    from candle_vllm_amd import tuning
    rng = np.random.default_rng(T + N + K)
    pair = mode == "silu"
    Nw = 2 * N if pair else N
    q, s = _gptq_case(rng, K, Nw, gs, dt)
    x = G.round_dt(rng.normal(0, 1, (T, K)), dt)
    z = rng.integers(1, 17, ((1 if gs == -1 else K // gs), Nw)) if mode == "zp" else None
    kw = {}
    if z is not None:
        kw = {"qzeros": dev_u32(G.gptq_pack_zeros(z)), "zero_mode": cv.ZERO_GPTQ_PLUS1}
    lin = cv.GPTQLinear(dev_u32(G.gptq_pack(q)), dev16(s, dt), gs if gs > 0 else K, **kw)
    assert lin.tiled
    lin_ref = G.gptq_linear(x, G.gptq_dequant(q, s, z, gs), None, dt)
    fkw, ulps, mag = {}, 1.01, None
    if pair:
        ref, fkw, ulps = G.silu_mul16(lin_ref[:, :N], lin_ref[:, N:], dt), {"epilogue": cv.EPI_SILU_MUL}, 3.0
    elif mode == "resid":
        res = G.round_dt(rng.normal(0, 1, (T, N)), dt)
        ref, fkw, ulps, mag = G.round_dt(lin_ref + res, dt), {"epilogue": cv.EPI_RESID, "residual": dev16(res, dt)}, 2.01, lin_ref
    else:
        ref = lin_ref
    y = host16(lin.forward(dev16(x, dt), **fkw), dt)
    check_ulp(y, ref, dt, ulps=ulps, what="gptq prompt GEMM", mag=mag)
    with tuning(30, 16):                                          # key 30 bit 4: the one-pass 4-bit prompt GEMM off
        y0 = host16(lin.forward(dev16(x, dt), **fkw), dt)
    check_ulp(y0, ref, dt, ulps=ulps, what="decode kernel in chunks", mag=mag)


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("T,N,K,gs,mode", [(32, 1184, 3584, 128, "store"), (32, 3584, 3584, 128, "resid"), (9, 224, 2048, 64, "store"),
                                           (48, 512, 4864, 256, "bias"), (17, 2 * 4736, 1792, 128, "silu"), (32, 2 * 608, 3584, 128, "silu"),
                                           (5, 64, 18944, 128, "resid"), (33, 4096, 256, 128, "store")])
def test_gptq_wide_kernel_5_to_48_tokens(cv, dt, T, N, K, gs, mode):
    """round 5 (csrc/gptq_wide.inc): 5..48 tokens over the tiled 4-bit image with the activations shared through LDS -- many-tile launches
    in one sweep (gate/up as pairs per wave, SiLU * up from the accumulators), few-tile launches split over K with the partial sums added
    by the epilogue launch (Qwen2-7B's wo / down shapes: 14 and 74 k-blocks), a single k-block, ragged last workgroups.  Against the oracle
    at the bound of the 16-token-tile kernel it replaces, and against that kernel itself (tuning key 30 bit 128)."""
    from candle_vllm_amd import tuning
    rng = np.random.default_rng(K + N + T)
    q, s = _gptq_case(rng, K, N, gs, dt)
    x = G.round_dt(rng.normal(0, 1, (T, K)), dt)
    bias = G.round_dt(rng.normal(0, 1, N), dt) if mode == "bias" else None
    lin = cv.GPTQLinear(dev_u32(G.gptq_pack(q)), dev16(s, dt), gs, bias=None if bias is None else dev16(bias, dt))
    assert lin.tiled
    full = G.gptq_linear(x, G.gptq_dequant(q, s, None, gs), bias, dt)
    kw, ulps = {}, 1.0
    if mode == "silu":
        ref = G.silu_mul16(full[:, :N // 2], full[:, N // 2:], dt)
        kw, ulps = dict(epilogue=cv.EPI_SILU_MUL), 3.0
    elif mode == "resid":
        res = G.round_dt(rng.normal(0, 1, (T, N)), dt)
        ref = G.round_dt(full + res, dt)
        kw, ulps = dict(epilogue=cv.EPI_RESID, residual=dev16(res, dt)), 2.01
    else:
        ref = full
        if mode == "bias":
            ulps = 2.01
    magkw = {"mag": full} if mode == "resid" else ({"mag": G.gptq_linear(x, G.gptq_dequant(q, s, None, gs), None, dt)} if mode == "bias" else {})
    y = host16(lin.forward(dev16(x, dt), **kw), dt)
    check_ulp(y, ref, dt, ulps=ulps, what="gptq wide kernel", **magkw)
    with tuning(30, 128):                                          # key 30 bit 128: the wide kernel off
        y0 = host16(lin.forward(dev16(x, dt), **kw), dt)
    check_ulp(y0, ref, dt, ulps=ulps, what="16-token-tile kernel", **magkw)
    assert np.abs(y - y0).max() <= 3 * (2.0 ** -8 if dt == "bf16" else 2.0 ** -11) * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("T,N,K,tiled", [(32, 256, 8192, True), (20, 336, 14336, True), (48, 64, 8448, False), (9, 4096, 8192, True)])
def test_wide_16bit_kernel_few_tiles_long_k_split(cv, dt, T, N, K, tiled):
    """round 5: few-tile 16-bit launches with a long K (down_proj of a batch-32 step: 256 tiles x 56 k-blocks at Llama-3-8B) take the
    LDS-shared-activation sweep split over K across workgroups; the partial sums meet in the epilogue launch (split order), which applies
    bias / residual and the rounding chain.  Against the oracle at the bound of the kernel it replaces, bias + residual included."""
    rng = np.random.default_rng(T + N + K)
    x = G.round_dt(rng.normal(0, 1, (T, K)), dt)
    w = G.round_dt(rng.normal(0, 0.02, (N, K)), dt)
    b = G.round_dt(rng.normal(0, 0.2, N), dt)
    res = G.round_dt(rng.normal(0, 1, (T, N)), dt)
    lin = cv.Linear(dev16(w, dt), dev16(b, dt), tiled=tiled)
    y = host16(lin.forward(dev16(x, dt), epilogue=cv.EPI_RESID, residual=dev16(res, dt)), dt)
    full = G.linear16(x, w, b, dt)
    check_ulp(y, G.round_dt(full + res, dt), dt, what="split-K wide kernel + bias + residual", ulps=3.01, mag=np.maximum(np.abs(full), np.abs(G.linear16(x, w, None, dt))))
    y2 = host16(cv.Linear(dev16(w, dt), tiled=tiled).forward(dev16(x, dt)), dt)
    check_ulp(y2, G.linear16(x, w, None, dt), dt, what="split-K wide kernel")
