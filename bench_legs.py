"""Secondary legs of bench.py: BASELINE.json configs[2..4] at their named shapes on ONE MI355X (the TP variants need the
driver's multi-GPU run), each with algorithmic bytes, achieved GB/s, roofline fraction and hipGraph replay.

  bf16_b32     configs[2]  Llama-3-8B bf16 safetensors host path, batch 32 ragged contexts U[256,4096]  (llama.rs:139-201)
  gptq_qwen2   configs[3]  Qwen2-7B GPTQ 4-bit g128 (marlin_4bit arm), batch 1, ctx 4096               (qwen.rs:78-96)
  mixtral_fp8  configs[4]  Mixtral-8x7B Q4_K GGUF, 8 experts top-2, fp8 KV cache, batch 1, ctx 4096    (quantized_llama.rs:56-123)
Synthetic weights of the named architectures (no network for checkpoints).  A "step" = one greedy decode step of the whole
model over the batch through the library's own step drivers (mi355_llama_decode_* / mi355_dense_decode_*: forward, argmax and
the device-side input advance in one captured hipGraph); every step reads its sampled tokens back, as the engine does."""
import os
import time
from types import SimpleNamespace

import numpy as np

HBM_PEAK_GBS = 8000.0


def _dims(**kw):
    d = dict(rms_eps=1e-5, max_seq=8192, block_size=64, qkv_bias=False, layer_norm=False, rotary_dim=0, kv_fp8=False)
    d.update(kw)
    return SimpleNamespace(**d)


def llama3_8b_dense():
    return _dims(hidden=4096, n_layers=32, n_heads=32, n_kv_heads=8, head_dim=128, intermediate=14336, vocab=128256, rope_theta=500000.0)


def qwen2_7b():
    """Qwen2-7B (public model card): hidden 3584, 28 layers, 28 heads, 4 kv heads, intermediate 18944, qkv bias"""
    return _dims(hidden=3584, n_layers=28, n_heads=28, n_kv_heads=4, head_dim=128, intermediate=18944, vocab=152064,
                 rope_theta=1000000.0, rms_eps=1e-6, qkv_bias=True)


def _result(name, workload, B, steps, tb, weight_bytes, kv_bytes, extra=None):
    """tb: bench_timing.timed_blocks result -- `value` is the MEDIAN block of `steps` steps; min / max / every block ride along"""
    dt = tb["median_s"]
    step_bytes = weight_bytes + kv_bytes
    gbs = step_bytes * steps / dt / 1e9
    r = {"config": name, "workload": workload, "value": round(B * steps / dt, 1), "unit": "tokens/s", "batch": B, "steps": steps,
         "ms_per_step": round(1e3 * dt / steps, 3), "algorithmic_bytes": int(step_bytes), "achieved_GBs": round(gbs, 1),
         "frac_of_8TBs": round(gbs / HBM_PEAK_GBS, 4), "roofline_tok_s_at_8TBs": round(B * HBM_PEAK_GBS * 1e9 / step_bytes, 1),
         "graph": True, "value_is": "median of %d blocks" % len(tb["blocks_ms_per_step"]),
         "value_min": round(B * steps / tb["max_s"], 1), "value_max": round(B * steps / tb["min_s"], 1),
         "blocks_ms_per_step": tb["blocks_ms_per_step"], "settle": tb["settle"],
         "graph_captures_in_timed_region": tb.get("graph_captures_in_timed_region"),
         "eager_steps_in_timed_region": tb.get("eager_steps_in_timed_region")}
    if extra:
        r.update(extra)
    return r


def _dense_graph_loop(gm, cfg, B, ctxs, steps, warmup, kv_elem):
    """greedy decode loop of the 16-bit host layer through the library's own step driver (mi355_dense_decode_begin / _step /
    _read_tokens: forward, argmax and the next-step inputs are ONE captured hipGraph, no torch op inside the step -- graph.rs:471-661,
    inputs.rs:389-423); every step reads its sampled tokens back, as the engine does"""
    import torch
    bs = cfg.block_size
    cap = int(max(ctxs)) + steps + warmup + 2
    nblk = [-(-(int(c) + steps + warmup + 2) // bs) for c in ctxs]
    maxb = max(nblk)
    rng = np.random.default_rng(0)
    ids = rng.permutation(sum(nblk))
    bt_h = np.zeros((B, maxb), np.uint32)
    o = 0
    for i, n in enumerate(nblk):
        bt_h[i, :n] = ids[o:o + n]
        o += n
    tok = rng.integers(0, cfg.vocab, B).astype(np.uint32)
    stream = torch.cuda.Stream()
    st = stream.cuda_stream
    gm.set_graph(True)
    import bench_timing

    def reset():
        gm.decode_begin(tok, np.asarray(ctxs, np.uint32), bt_h, ctx_cap=cap, stream=st)

    def step():
        gm.decode_step(st)
        gm.read_tokens(st)                                            # sampled tokens -> host every step
    # settle by time (the first step of the shape is eager, the second captures, the rest replay), then 3 blocks of `steps` steps, each
    # from the same start state behind `warmup` untimed steps
    tb = bench_timing.timed_blocks(step, torch.cuda.synchronize, steps, warmup=warmup, blocks=3, reset=reset, stats=gm.graph_stats)
    mean_ctx = float(np.mean(ctxs)) + 1 + warmup + (steps - 1) / 2.0
    kv = B * (mean_ctx + 1) * 2 * cfg.n_layers * cfg.n_kv_heads * cfg.head_dim * kv_elem
    return tb, kv


def leg_bf16_b32(steps=16, warmup=3):
    from candle_vllm_amd import dense_model as DM
    cfg = llama3_8b_dense()
    B = 32
    ctxs = np.random.default_rng(4321).integers(256, 4097, B).tolist()
    gm = DM.DenseLlama(cfg, max_batch=B, max_blocks_per_seq=80, kv_layout=DM.KV_PAGED)
    gm.load_synthetic()
    gm.alloc_kv_cache(B * 70 + 8)
    tb, kv = _dense_graph_loop(gm, cfg, B, ctxs, steps, warmup, 2)
    return _result("bf16_b32", "BASELINE configs[2]: Llama-3-8B bf16, batch 32, ragged contexts U[256,4096] in paged KV (block 64), "
                   "16-bit host layer (dense_model.cpp)", B, steps, tb, gm.weight_bytes(), kv)


def leg_bf16_prompt(T=2048):
    """one prompt step of T tokens through the 16-bit host layer (dense_model.cpp): Llama-3-8B bf16, every projection on the
    hand-written 128 x 128 MFMA GEMM (dense_gemv.hip dense_gemm_kernel), prefill attention, cache write -- tokens/s and the useful
    TFLOP/s of the weight GEMMs (2 * parameters * T; attention's flops not counted)"""
    import torch
    from candle_vllm_amd import dense_model as DM
    from candle_vllm_amd.block_engine import BlockEngine
    cfg = llama3_8b_dense()
    nblk = -(-T // cfg.block_size)
    gm = DM.DenseLlama(cfg, max_batch=1, max_blocks_per_seq=nblk + 1, kv_layout=DM.KV_PAGED)
    gm.load_synthetic()
    gm.alloc_kv_cache(nblk + 2)
    eng = BlockEngine(cfg.block_size, nblk + 1, 0)
    seq = eng.new_sequence(0, np.random.default_rng(99).integers(0, cfg.vocab, T).tolist())
    eng.allocate([seq])
    meta = eng.prepare_prompt([seq])
    import bench_timing
    tc = bench_timing.timed_calls(lambda: gm.forward(meta, is_prefill=True, sync=False), torch.cuda.synchronize)
    dt = tc["median_s"]
    # useful flops from the SHAPES: 2 x projection weights x T; the lm_head runs on the last row only (x 1 token); the embedding is a gather
    H, Hkv, D, hid, I = cfg.n_heads, cfg.n_kv_heads, cfg.head_dim, cfg.hidden, cfg.intermediate
    layer_params = 2 * hid * H * D + 2 * hid * Hkv * D + 3 * hid * I
    gemm_flops = 2.0 * (cfg.n_layers * layer_params * T + cfg.vocab * hid * 1)
    attn_flops = cfg.n_layers * 2.0 * 2.0 * (T * (T + 1) / 2.0) * H * D
    useful = gemm_flops / dt / 1e12
    return {"config": "bf16_prompt", "workload": f"Llama-3-8B bf16, one prompt step of {T} tokens (16-bit host layer, own MFMA GEMM, no library GEMM)",
            "value": round(T / dt, 1), "unit": "prompt tokens/s", "tokens": T, "ms": round(dt * 1e3, 2), "useful_TFLOPs": round(useful, 1),
            "value_is": "median of %d calls" % len(tc["calls_ms"]), "value_min": round(T / tc["max_s"], 1), "value_max": round(T / tc["min_s"], 1),
            "calls_ms": tc["calls_ms"], "settle": tc["settle"],
            "frac_of_2.5PF_dense_bf16": round(useful / 2500.0, 3),
            "with_causal_attention_TFLOPs": round((gemm_flops + attn_flops) / dt / 1e12, 1)}


def leg_gptq_qwen2(steps=32, warmup=4, B=1):
    """Qwen2-7B GPTQ 4-bit, group 128: every projection through the marlin_4bit arm (checkpoint layout, Marlin-permuted scales
    un-permuted by index arithmetic), lm_head / embedding 16-bit"""
    import torch
    from candle_vllm_amd import dense_model as DM
    cfg = qwen2_7b()
    gm = DM.DenseLlama(cfg, max_batch=B, max_blocks_per_seq=80, kv_layout=DM.KV_PAGED)
    g = torch.Generator(device="cuda").manual_seed(3)
    rng = np.random.default_rng(3)

    def dev16(n, s=0.02, mean=0.0):
        return (torch.randn(n, generator=g, device="cuda") * s + mean).to(torch.bfloat16)

    def put(layer, name, t):
        DM._check(DM.lib.mi355_dense_set_weight_dev(gm.h, layer, DM.W_SLOTS[name], t.data_ptr(), t.numel()), name)
        torch.cuda.synchronize()
    HD, KD, hid, I, group = cfg.n_heads * cfg.head_dim, cfg.n_kv_heads * cfg.head_dim, cfg.hidden, cfg.intermediate, 128
    put(-1, "tok_embd", dev16(cfg.vocab * hid, 0.5)); put(-1, "output_norm", dev16(hid, 0.02, 1.0)); put(-1, "output", dev16(cfg.vocab * hid))
    shapes = {"wq": (HD, hid), "wk": (KD, hid), "wv": (KD, hid), "wo": (hid, HD), "w1": (I, hid), "w3": (I, hid), "w2": (hid, I)}
    packs = {}
    wbytes = 2 * cfg.vocab * hid
    for name, (n, k) in shapes.items():                               # one random packed tensor per shape, shared by the layers
        if (n, k) not in packs:
            packs[(n, k)] = (rng.integers(0, 2 ** 32, (k // 8, n), dtype=np.uint32),
                             rng.uniform(0.002, 0.01, (k // group, n)).astype(np.float32))
    for l in range(cfg.n_layers):
        put(l, "attn_norm", dev16(hid, 0.02, 1.0)); put(l, "ffn_norm", dev16(hid, 0.02, 1.0))
        put(l, "bq", dev16(HD)); put(l, "bk", dev16(KD)); put(l, "bv", dev16(KD))
        for name, (n, k) in shapes.items():
            qw, sc = packs[(n, k)]
            gm.set_gptq(l, name, qw, sc, group)
            wbytes += (k // 8) * n * 4 + (k // group) * n * 2
    ctxs = [4096] if B == 1 else np.random.default_rng(4321).integers(256, 4097, B).tolist()
    gm.alloc_kv_cache(sum(-(-(int(c) + steps + warmup + 2) // cfg.block_size) for c in ctxs) + 8)
    tb, kv = _dense_graph_loop(gm, cfg, B, ctxs, steps, warmup, 2)
    if B > 1:
        return _result(f"gptq_qwen2_b{B}", f"BASELINE configs[3] shapes at batch {B}: Qwen2-7B GPTQ 4-bit (group 128, marlin_4bit arm), ragged "
                       "contexts U[256,4096] in paged KV (block 64): the 5..64-token kernels over the tiled 4-bit image", B, steps, tb, wbytes, kv)
    return _result("gptq_qwen2", "BASELINE configs[3] on one GPU: Qwen2-7B GPTQ 4-bit (group 128, marlin_4bit arm), batch 1, ctx 4096 in "
                   "paged KV (block 64); TP=2 needs the driver's multi-GPU run", 1, steps, tb, wbytes, kv)


def leg_gptq_qwen2_b32(steps=12, warmup=3):
    return leg_gptq_qwen2(steps=steps, warmup=warmup, B=32)


def _build_mixtral(B, bps, max_seq=8192, n_layers=32):
    """Mixtral-8x7B Q4_K GGUF shapes on the device: device router + top-2 + experts, fp8 (e4m3fn) KV cache layout; one random expert per
    projection (native GGUF blocks on the host) shared by all experts and layers -- the loader repacks and uploads every (layer, expert)
    copy, so the model is full size on the device.  Returns (model, cfg, weight bytes a single token touches per step, rng, generator)."""
    import torch
    from candle_vllm_amd import model as M
    cfg = M.ModelDims(hidden=4096, n_layers=n_layers, n_heads=32, n_kv_heads=8, head_dim=128, intermediate=14336, vocab=32000,
                      rope_theta=1000000.0, max_seq=max_seq, block_size=64)
    cfg.n_expert, cfg.n_expert_used = 8, 2
    gm = M.GGUFLLaMa(cfg, max_batch=B, max_blocks_per_seq=bps, kv_layout=M.KV_PAGED_FP8)
    lib = M.lib
    gen = torch.Generator(device="cuda").manual_seed(5)
    rng = np.random.default_rng(5)
    hid, I, H, Hkv, D = cfg.hidden, cfg.intermediate, cfg.n_heads, cfg.n_kv_heads, cfg.head_dim

    def f32(layer, which, a):
        a = np.ascontiguousarray(a, np.float32)
        M._check(lib.mi355_llama_set_f32(gm.h, layer, which, a.ctypes.data, a.size), "set_f32")

    def tiles(layer, which, n, k, t=M.GGML_Q4_K):
        tl = M.random_tiles(t, n, k, "cuda", gen)
        gm._keep.append(tl)
        M._check(lib.mi355_llama_set_qweight_tiles(gm.h, layer, which, t, tl.data_ptr(), n, k), "set_tiles")
        return (n // 16) * (k // 256) * (2304 if t == M.GGML_Q4_K else 3360)
    emb = (torch.randn((cfg.vocab, hid), device="cuda", generator=gen) * 0.02).cpu().numpy()
    f32(-1, M.W_TOK_EMBD, emb)
    f32(-1, M.W_OUTPUT_NORM, 1.0 + rng.normal(0, 0.02, hid))
    step_w = tiles(-1, M.W_OUTPUT, cfg.vocab, hid, M.GGML_Q6_K)

    def native_q4k(n, k):
        b = rng.integers(0, 256, (n, k // 256, 144), dtype=np.uint8)
        b[:, :, 0:2] = np.array([2e-4], np.float16).view(np.uint8)
        b[:, :, 2:4] = np.array([1.5e-3], np.float16).view(np.uint8)
        return np.ascontiguousarray(b)
    e_up, e_down = native_q4k(I, hid), native_q4k(hid, I)
    for l in range(cfg.n_layers):
        f32(l, M.W_ATTN_NORM, 1.0 + rng.normal(0, 0.02, hid))
        f32(l, M.W_FFN_NORM, 1.0 + rng.normal(0, 0.02, hid))
        f32(l, 12, rng.normal(0.0, 0.5, (cfg.n_expert, hid)))
        step_w += tiles(l, M.W_WQ, H * D, hid) + tiles(l, M.W_WK, Hkv * D, hid) + tiles(l, M.W_WV, Hkv * D, hid) + tiles(l, M.W_WO, hid, H * D)
        for e in range(cfg.n_expert):
            for which, blk, n, k in ((M.W_W1, e_up, I, hid), (M.W_W3, e_up, I, hid), (M.W_W2, e_down, hid, I)):
                M._check(lib.mi355_llama_set_moe_expert(gm.h, l, which, e, M.GGML_Q4_K, blk.ctypes.data, n, k), "set_moe_expert")
        step_w += cfg.n_expert_used * 3 * I * (hid // 256) * 144      # a token touches top-2 of the 8 experts
    return gm, cfg, step_w, rng, gen


def leg_mixtral_fp8(steps=32, warmup=4, B=1):
    """Mixtral-8x7B Q4_K GGUF shapes: device router + top-2 + expert mat-vecs + combine, fp8 (e4m3fn) KV cache"""
    import torch
    from candle_vllm_amd import model as M
    K, Wm = steps, warmup
    bps = -(-(4096 + K + Wm + 2) // 64)
    gm, cfg, step_w, rng, gen = _build_mixtral(B, bps)
    lib = M.lib
    hid, I, H, Hkv, D = cfg.hidden, cfg.intermediate, cfg.n_heads, cfg.n_kv_heads, cfg.head_dim
    ctxs = [4097] if B == 1 else np.random.default_rng(4321).integers(256, 4097, B).tolist()       # B > 1: ragged contexts as the bf16 leg
    nblk = [-(-(int(c_) + K + Wm + 2) // cfg.block_size) for c_ in ctxs]
    num_blocks = sum(nblk) + 8
    gm.alloc_kv_cache(num_blocks)
    n8 = lib.mi355_llama_kv_bytes_per_tensor(gm.h)
    for l in range(cfg.n_layers):                                     # random e4m3 bytes (finite codes only)
        for which in (0, 1):
            t = torch.randint(0, 120, (n8,), dtype=torch.uint8, device="cuda", generator=gen)
            M._check(lib.mi355_llama_kv_copy(gm.h, l, which, t.data_ptr(), n8, 1), "kv_copy")
    perm = rng.permutation(num_blocks - 1) + 1
    bt = np.zeros((B, bps), np.uint32)
    o = 0
    for i, n in enumerate(nblk):
        bt[i, :n] = perm[o:o + n]
        o += n
    stream = torch.cuda.Stream()
    st = stream.cuda_stream
    gm.set_graph(True)
    import bench_timing
    tok0 = rng.integers(0, cfg.vocab, B).astype(np.uint32)

    def reset():
        gm.decode_begin(tok0, np.asarray(ctxs, np.uint32), bt, ctx_cap=4096 + K + Wm + 2, stream=st)

    def step():
        gm.decode_step(st); gm.read_tokens(st)
    tb = bench_timing.timed_blocks(step, torch.cuda.synchronize, K, warmup=Wm, blocks=3, reset=reset, stats=gm.graph_stats)
    mean_ctx = float(np.mean(ctxs)) + Wm + (K - 1) / 2.0
    kv = B * (mean_ctx + 1) * 2 * cfg.n_layers * Hkv * D * 1        # fp8: one byte per element
    if B > 1:
        # algorithmic bytes of a batch: every expert that ANY token selects is read once; with 2 * B >= 64 draws over 8 experts that is
        # all 8 in practice -- counted as all 8 (the per-pair path reads 2 * B experts per layer, the grouped one 8 x ceil(2B / 32))
        step_w += cfg.n_layers * (cfg.n_expert - cfg.n_expert_used) * 3 * I * (hid // 256) * 144
        return _result(f"mixtral_fp8_b{B}", f"Mixtral-8x7B Q4_K GGUF shapes (8 experts, top-2, device router), fp8 e4m3 KV cache, batch {B}, ragged "
                       "contexts U[256,4096]; (token, slot) pairs grouped by expert on the device (graph-safe), every expert streamed once per "
                       "32-row chunk of its block", B, K, tb, step_w, kv)
    return _result("mixtral_fp8", "BASELINE configs[4] on one GPU: Mixtral-8x7B Q4_K GGUF shapes (8 experts, top-2, device router), fp8 "
                   "e4m3 KV cache, batch 1, ctx 4096; TP=8 needs the driver's multi-GPU run", 1, K, tb, step_w, kv)


def leg_mixtral_fp8_b32(steps=16, warmup=3):
    return leg_mixtral_fp8(steps=steps, warmup=warmup, B=32)


def leg_engine_b32(n_req=32, new_lo=48, new_hi=80, parity_reqs=2):
    """BASELINE configs[2]'s "continuous batching ... staggered arrival" THROUGH the engine at Llama-3-8B Q4_K_M size: 32 requests
    with prompts U[256,4096] arrive four every three engine steps; the C++ scheduler (mi355_sched_schedule) admits prompt steps between
    decode steps, the block manager builds every step's inputs (mi355_be_prepare_*), decode steps replay the step driver's hipGraph of
    the current batch size, finished requests free their blocks -- candle_vllm_amd/engine.py.  Reported the reference's way
    (llm_engine.rs:984-1002: mean per-request decode tok/s x requests) beside the wall-clock rates and the fixed-batch number."""
    import torch
    from candle_vllm_amd import model as M
    from candle_vllm_amd import block_engine as be
    from candle_vllm_amd import engine as E
    cfg = M.ModelDims.llama3_8b()
    rng = np.random.default_rng(2468)
    plens = rng.integers(256, 4097, n_req)
    n_new = rng.integers(new_lo, new_hi + 1, n_req)
    bps = -(-(4096 + new_hi + 2) // cfg.block_size)
    nblk = int(sum(-(-(int(p) + int(n) + 1) // cfg.block_size) for p, n in zip(plens, n_new))) + 64
    gm = M.GGUFLLaMa(cfg, max_batch=n_req, max_blocks_per_seq=bps, kv_layout=M.KV_PAGED)
    gm.load_synthetic(seed=1235, recipe="q4_k_m", scale=0.2)      # branch gain < 1 as in a trained checkpoint: logits can be compared (timing is value-independent)
    gm.alloc_kv_cache(nblk)
    prompts = [rng.integers(0, cfg.vocab, int(p)).tolist() for p in plens]

    def make(ids):
        return [E.Request(i, prompts[i], int(n_new[i]), arrival_step=3 * (k // 4)) for k, i in enumerate(ids)]

    def sched():
        return be.Scheduler(block_size=cfg.block_size, num_gpu_blocks=nblk, num_cpu_blocks=0, max_num_parallel_reqs=n_req,
                            max_num_batched_tokens=8192, prefill_chunk_size=0)
    stream = torch.cuda.Stream()
    # warm-up run: every batch size's graph and every prompt shape's workspace exist before the timed run, as after the reference's
    # start-up capture (graph.rs:471-661 captures batch sizes 1..max at load)
    warm = make(range(n_req))
    E.run_engine(gm, sched(), warm, stream=stream.cuda_stream, graph=True)
    torch.cuda.synchronize()
    runs = []
    for _ in range(3):                                                # three timed runs of the whole request set; the median one is reported
        reqs_i = make(range(n_req))
        st_i = E.run_engine(gm, sched(), reqs_i, stream=stream.cuda_stream, graph=True)
        torch.cuda.synchronize()
        runs.append((E.usage_summary(reqs_i)["decode_throughput"], reqs_i, st_i))
    runs.sort(key=lambda t: t[0])
    _, reqs, st = runs[1]
    u = E.usage_summary(reqs)
    dec_tokens = u["completion_tokens"] - n_req
    same_as_warm = sum(a.tokens == b.tokens for a, b in zip(warm, reqs))
    r = {"config": "engine_b32", "workload": f"BASELINE configs[2] through the engine: Llama-3-8B Q4_K_M, {n_req} requests, prompts U[256,4096], "
         f"{new_lo}..{new_hi} new tokens each, arrivals staggered 4 per 3 engine steps; scheduler -> block manager -> prepare_* -> model -> token",
         "value": round(u["decode_throughput"], 1), "unit": "tokens/s (reference definition: mean per-request decode tok/s x requests, llm_engine.rs:984-1002)",
         "decode_tps_avg_per_request": round(u["decode_tps_avg"], 2), "prompt_throughput_ref_definition": round(u["prompt_throughput"], 1),
         "wall_s": round(st["wall_s"], 3), "wall_tokens_per_s_all": round((u["prompt_tokens"] + u["completion_tokens"]) / st["wall_s"], 1),
         "wall_generated_tokens_per_s": round(u["completion_tokens"] / st["wall_s"], 1),
         "requests": n_req, "prompt_tokens": u["prompt_tokens"], "completion_tokens": u["completion_tokens"],
         "prompt_steps": st["prompt_steps"], "decode_steps": st["decode_steps"], "max_batch": st["max_batch"], "preempted": st["preempted"],
         "mean_decode_batch": round(dec_tokens / max(st["decode_steps"], 1), 2), "graph": True,
         "run_to_run_identical_requests": int(same_as_warm), "value_is": "median of 3 runs of the request set (after one warm-up run)",
         "value_min": round(runs[0][0], 1), "value_max": round(runs[2][0], 1)}
    if parity_reqs:
        # batching invariance at full size: a third, untimed engine run keeps the logits row behind every token of two requests (the
        # shortest and the longest prompt); the SAME model then runs each of them alone (its own scheduler and blocks, batch 1: the
        # 1..8-token kernels, eager steps), teacher-forced with the engine's tokens.  The two runs differ in every kernel of the decode
        # path (9..32-token GEMMs with one f16 activation plane and the balanced attention stream against the single-token mat-vecs and the
        # one-partition attention waves): the logits must agree within what those paths are held to against the oracle at this size
        # (2e-2 of the logit scale through 32 layers at batch 32), and a token may differ only inside that band.
        ids = [int(np.argmin(plens)), int(np.argmax(plens))][:parity_reqs]
        traced = make(range(n_req))
        E.run_engine(gm, sched(), traced, stream=stream.cuda_stream, graph=True, trace_ids=set(ids))
        worst, flips, hard, steps = 0.0, 0, 0, 0
        same_tokens = all(traced[i].tokens == reqs[i].tokens for i in range(n_req))
        for rid in ids:
            s1 = be.Scheduler(block_size=cfg.block_size, num_gpu_blocks=nblk, num_cpu_blocks=0, max_num_parallel_reqs=1,
                              max_num_batched_tokens=8192, prefill_chunk_size=0)
            eng = s1.block_engine
            seq = eng.new_sequence(0, prompts[rid])
            s1.add_sequence(0, [seq])
            s1.schedule()
            lg = gm.forward_prefill(eng.prepare_prompt([seq]))
            for step, tok in enumerate(traced[rid].tokens):
                alone = lg[0].cpu().numpy()
                got = traced[rid].logits[step]
                err = float(np.abs(alone - got).max() / np.abs(alone).max())
                worst = max(worst, err)
                steps += 1
                if int(alone.argmax()) != tok:
                    if alone.max() - alone[tok] <= 2.0 * err * np.abs(alone).max() + 1e-12:
                        flips += 1
                    else:
                        hard += 1
                seq.add_token(int(tok))                                # teacher-forced: the engine's token
                s1.schedule()
                if step + 1 < len(traced[rid].tokens):
                    lg = gm.forward_decode(eng.prepare_decode([seq]))
        r["parity"] = {"kind": "batching invariance: engine run (ragged batches, 9..32-token kernels) vs the same model on the request alone "
                               "(batch 1 kernels), teacher-forced with the engine's tokens", "requests_checked": len(ids), "steps_compared": steps,
                       "logits_max_rel_err": round(worst, 6), "tokens_differing_inside_the_error_band": flips, "tokens_differing_outside_it": hard,
                       "traced_run_tokens_equal_timed_run": bool(same_tokens), "oracle": "none (self-consistency; the kernels' oracle parity: parity.batch32)"}
    return r


def leg_mixtral_prompt_16k(T=16384, chunk=8192):
    """BASELINE configs[4] at its stated size on one GPU: Mixtral-8x7B Q4_K, fp8 KV cache, a 16 384-token prompt
    prefilled in two 8192-token chunks (`prefill_chunk_size`, pipelines/inputs.rs:90-230, main.rs:635-638): the scheduler admits the
    first chunk, re-queues the sequence, the second chunk attends over the first through the e4m3 cache (cached prefix) -- prompt tokens/s
    over both steps.  Full-size parity of this path: tests/test_gpu_fullsize.py::test_mixtral_chunked_prefill_16k_one_layer_full_width."""
    import torch
    from candle_vllm_amd import block_engine as be
    from candle_vllm_amd import ops as cvo
    bps = -(-(T + 8) // 64)
    gm, cfg, step_w, rng, gen = _build_mixtral(1, bps, max_seq=T + 64)
    gm.alloc_kv_cache(bps + 4)
    prompt = rng.integers(0, cfg.vocab, T).tolist()

    def one_pass():
        sched = be.Scheduler(block_size=cfg.block_size, num_gpu_blocks=bps + 4, num_cpu_blocks=0, max_num_parallel_reqs=1,
                             max_num_batched_tokens=chunk, prefill_chunk_size=chunk)
        eng = sched.block_engine
        seq = eng.new_sequence(0, prompt)
        sched.add_sequence(0, [seq])
        steps, tok = [], None
        while True:
            out = sched.schedule()
            assert out.is_prompt and out.scheduled == [0]
            meta = eng.prepare_prompt([seq], chunk=chunk)
            t0 = time.perf_counter()
            lg = gm.forward_prefill(meta)
            torch.cuda.synchronize()
            steps.append((len(meta["input_ids"]), time.perf_counter() - t0))
            if sched.filter_prefill_finished(out.scheduled):
                tok = int(cvo.argmax(lg)[0])
                break
        return steps, tok
    one_pass()                                                        # warm-up: workspaces of both chunk shapes
    passes = sorted((one_pass() for _ in range(3)), key=lambda p_: sum(t for _, t in p_[0]))      # three timed passes, the median one is reported
    steps, tok = passes[1]
    dt = sum(t for _, t in steps)
    flops = 2.0 * (step_w - cfg.vocab * (cfg.hidden // 256) * 210) / 0.5625 * T   # projections + top-2 experts per token (Q4_K: 0.5625 B / weight); lm_head runs once
    return {"config": "mixtral_prompt_16k", "workload": f"BASELINE configs[4] at size: Mixtral-8x7B Q4_K shapes, fp8 e4m3 KV cache, {T}-token prompt "
            f"as {len(steps)} chunks of {chunk} tokens through the scheduler (cached-prefix prompt attention over the fp8 cache, experts grouped per chunk)",
            "value": round(T / dt, 1), "unit": "prompt tokens/s", "tokens": T, "chunks": [[n, round(t * 1e3, 1)] for n, t in steps],
            "ms": round(dt * 1e3, 1), "useful_TFLOPs": round(flops / dt / 1e12, 1), "first_token": tok,
            "value_is": "median of 3 passes (after one warm-up pass)",
            "value_min": round(T / sum(t for _, t in passes[2][0]), 1), "value_max": round(T / sum(t for _, t in passes[0][0]), 1)}


LEGS = {"bf16_b32": leg_bf16_b32, "gptq_qwen2": leg_gptq_qwen2, "mixtral_fp8": leg_mixtral_fp8, "bf16_prompt": leg_bf16_prompt, "mixtral_fp8_b32": leg_mixtral_fp8_b32, "gptq_qwen2_b32": leg_gptq_qwen2_b32, "engine_b32": leg_engine_b32, "mixtral_prompt_16k": leg_mixtral_prompt_16k}
NO_PARITY_LEG = {"bf16_prompt", "engine_b32", "mixtral_prompt_16k"}    # bf16_prompt: test_gpu_linear.py (>= 96 tokens), test_gpu_dense_model.py; the other two carry / name their own parity   # covered by tests: test_gpu_linear.py (>= 96 tokens), test_gpu_dense_model.py; test_gpu_model.py (device-grouped experts)


def leg_parity(name):
    """The leg's geometry at its benchmarked size against the oracle (tests/fullsize_dense.py, tests/fullsize_moe.py: every
    layer teacher-forced from the oracle's stream + one / two greedy steps end to end).  The oracle is the CHECKER here, run
    after the timed region; PARITY UNPINNED (no reference-held vector exists for the float kernels)."""
    if name == "mixtral_fp8":
        from tests.fullsize_moe import MoePair
        p = MoePair(n_layers=32, scale=0.2)
        r = p.run(ctx=4097, steps=2)
    elif name == "mixtral_fp8_b32":
        from tests.fullsize_moe import MoePair
        from tests.fullsize_dense import ragged_batch32
        p = MoePair(n_layers=32, scale=0.2, max_batch=32, num_blocks=320)
        r = p.run_batch(ragged_batch32(np.random.default_rng(4321)), steps=1)      # asserted with per-layer + two steps: tests/test_gpu_fullsize.py
    else:
        from tests.fullsize_dense import DensePair, ragged_batch32
        base = "gptq_qwen2" if name.startswith("gptq_qwen2") else name
        p = DensePair(base, std=0.008 if base == "bf16_b32" else 0.004, max_batch=32 if name == "gptq_qwen2_b32" else None)
        r = p.run(ragged_batch32(np.random.default_rng(4321)) if name in ("bf16_b32", "gptq_qwen2_b32") else [4097])
    del p
    return {k: (round(v, 7) if isinstance(v, float) else v) for k, v in r.items() if k not in ("per_layer", "units")}


def run_legs(names, parity=True):
    import gc
    import torch
    out = {}
    for n in names:
        t0 = time.time()
        try:
            out[n] = LEGS[n]()
            out[n]["seconds"] = round(time.time() - t0, 1)
        except Exception as e:                                        # a secondary leg must never sink the headline
            out[n] = {"error": repr(e)}
        gc.collect()
        torch.cuda.empty_cache()
        if parity and "error" not in out[n] and n not in NO_PARITY_LEG:
            t1 = time.time()
            try:
                out[n]["parity"] = leg_parity(n)
                out[n]["parity"]["seconds"] = round(time.time() - t1, 1)
            except Exception as e:                                    # the checker must never sink the measured number
                out[n]["parity"] = {"error": repr(e)}
            gc.collect()
            torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    import json
    import sys
    args = sys.argv[1:]
    for kv in os.environ.get("MI355_TUNING", "").split(","):          # experiments: MI355_TUNING=key:value,key:value
        if ":" in kv:
            from candle_vllm_amd import lib as _lib
            _lib.mi355_set_tuning(int(kv.split(":")[0]), int(kv.split(":")[1]))
    parity = "--no-parity" not in args
    names = [a for a in args if not a.startswith("--")] or list(LEGS)
    print(json.dumps(run_legs(names, parity=parity)), flush=True)
