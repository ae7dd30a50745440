"""Full-size parity leg of the Mixtral configuration (TEST INFRASTRUCTURE: tests/test_gpu_fullsize.py, bench_legs.py's `parity`).

BASELINE configs[4] on one GPU, the geometry bench_legs.py times: Mixtral-8x7B Q4_K GGUF shapes -- hidden 4096, 32 layers, 32 / 8
heads, 8 experts of 14336 x 4096 with top-2 routing on the device, Q6_K lm_head -- with the fp8 (e4m3fn) KV cache, batch 1 at
context 4097 (MlpOrMoe::forward quantized_llama.rs:56-123, layers/moe.rs:746-810; is_fp8_keys attention.rs:574,896;
cache_engine.rs:304-311).  Oracle: oracle/llama.py's OracleLlama (router, top-k, per-token experts, fp8 cache read-back) with its
mat-vecs taken by the C twin's O1 product (oracle.c orc_qmatmul: dequantise, f64 dot) so a 32-layer step takes seconds.
Weights: native GGUF blocks built once per projection SHAPE and shared by layers and experts (as bench_legs.py does; the device
holds every (layer, expert) copy, so the model is full size there); routers and norms differ per layer, so different experts
are selected from layer to layer.  Checks: every layer teacher-forced from the oracle's stream (routing ids equal, stream after
the layer), then greedy steps end to end.   PARITY UNPINNED."""
import ctypes
import time

import numpy as np

from oracle import cref
from oracle import llama as OL
from oracle import ops as O


def _native(rng, ggml_type, n, k, scale):
    """random native GGUF blocks [n, k/256, block bytes] with sane f16 super-block scales"""
    if ggml_type == 12:                                                   # Q4_K: d, dmin, scales[12], qs[128]
        b = rng.integers(0, 256, (n, k // 256, 144), dtype=np.uint8)
        b[:, :, 0:2] = np.array([2e-4 * scale], np.float16).view(np.uint8)
        b[:, :, 2:4] = np.array([1.5e-3 * scale], np.float16).view(np.uint8)
    else:                                                                 # Q6_K: ql[128], qh[64], scales[16] i8, d
        b = rng.integers(0, 256, (n, k // 256, 210), dtype=np.uint8)
        b[:, :, 192:208] = rng.integers(-32, 32, (n, k // 256, 16), dtype=np.int8).view(np.uint8)
        b[:, :, 208:210] = np.array([1.2e-4 * scale], np.float16).view(np.uint8)
    return np.ascontiguousarray(b)


class MoePair:
    def __init__(self, n_layers=32, scale=1.0, seed=31, log=None, max_seq=8192, ctx_tokens=4096 + 16, max_batch=1, num_blocks=None):
        import torch
        from candle_vllm_amd import model as M
        self.torch, self.M = torch, M
        self.log = log or (lambda *a: None)
        lib = M.lib
        cfg = OL.LlamaConfig(vocab=32000, hidden=4096, n_layers=n_layers, n_heads=32, n_kv_heads=8, head_dim=128, intermediate=14336,
                             rms_eps=1e-5, rope_theta=1000000.0, max_seq=max_seq, block_size=64)
        self.cfg = cfg
        gcfg = M.ModelDims(hidden=4096, n_layers=n_layers, n_heads=32, n_kv_heads=8, head_dim=128, intermediate=14336, vocab=32000,
                           rope_theta=1000000.0, max_seq=max_seq, block_size=64)
        gcfg.n_expert, gcfg.n_expert_used = 8, 2
        rng = np.random.default_rng(seed)
        t0 = time.time()
        self.bps = -(-ctx_tokens // cfg.block_size)
        gm = M.GGUFLLaMa(gcfg, max_batch=max_batch, max_blocks_per_seq=self.bps, kv_layout=M.KV_PAGED_FP8)
        self.gm = gm
        hid, I, H, Hkv, D = cfg.hidden, cfg.intermediate, cfg.n_heads, cfg.n_kv_heads, cfg.head_dim

        def f32(layer, which, a):
            a = np.ascontiguousarray(a, np.float32)
            M._check(lib.mi355_llama_set_f32(gm.h, layer, which, a.ctypes.data, a.size), "set_f32")
            return a

        def qw(layer, which, tw):
            t, b = tw
            M._check(lib.mi355_llama_set_qweight(gm.h, layer, which, t, b.ctypes.data, b.shape[0], b.shape[1] * 256), "set_qweight")
        base = (rng.standard_normal((1000, hid)) * 0.02).astype(np.float32)
        W = {"tok_embd": f32(-1, M.W_TOK_EMBD, np.tile(base, (32, 1))[: cfg.vocab]),
             "output_norm": f32(-1, M.W_OUTPUT_NORM, 1.0 + rng.normal(0, 0.02, hid)),
             "output": (14, _native(rng, 14, cfg.vocab, hid, scale)), "layers": []}
        qw(-1, M.W_OUTPUT, W["output"])
        shared = {"wq": (12, _native(rng, 12, H * D, hid, scale)), "wk": (12, _native(rng, 12, Hkv * D, hid, scale)),
                  "wv": (12, _native(rng, 12, Hkv * D, hid, scale)), "wo": (12, _native(rng, 12, hid, H * D, scale))}
        e_w1, e_w3, e_w2 = (12, _native(rng, 12, I, hid, scale)), (12, _native(rng, 12, I, hid, scale)), (12, _native(rng, 12, hid, I, scale))
        slots = {"wq": M.W_WQ, "wk": M.W_WK, "wv": M.W_WV, "wo": M.W_WO}
        for l in range(n_layers):
            lw = {"attn_norm": f32(l, M.W_ATTN_NORM, 1.0 + rng.normal(0, 0.02, hid)),
                  "ffn_norm": f32(l, M.W_FFN_NORM, 1.0 + rng.normal(0, 0.02, hid)),
                  "gate_inp": f32(l, 12, rng.normal(0.0, 0.5, (8, hid)))}
            for name, which in slots.items():
                lw[name] = shared[name]
                qw(l, which, shared[name])
            lw["experts"] = [{"w1": e_w1, "w2": e_w2, "w3": e_w3} for _ in range(8)]
            for e in range(8):
                for which, tw, n, k in ((M.W_W1, e_w1, I, hid), (M.W_W3, e_w3, I, hid), (M.W_W2, e_w2, hid, I)):
                    M._check(lib.mi355_llama_set_moe_expert(gm.h, l, which, e, 12, tw[1].ctypes.data, n, k), "set_moe_expert")
            W["layers"].append(lw)
        self.W = W
        cref.build()
        self._fast_qmm = lambda x, tw, o2: cref.qmatmul(np.ascontiguousarray(x, np.float32), tw[1], tw[0], 0)   # the C twin's O1 product
        self.orc = OL.OracleLlama(cfg, W, flash_layout=False)
        self.orc.kv_fp8 = True
        self.log(f"mixtral: weights in both models: {time.time() - t0:.1f}s")
        t0 = time.time()
        self.num_blocks = num_blocks or (self.bps + 8)
        gm.alloc_kv_cache(self.num_blocks)
        ks, vs = O.kv_cache_shapes(self.num_blocks, cfg.block_size, Hkv, D, 1, False)
        kb = rng.integers(0, 120, ks, dtype=np.uint8)                     # finite e4m3 codes
        vb = rng.integers(0, 120, vs, dtype=np.uint8)
        self.cache = []
        for l in range(n_layers):
            k, v = np.roll(kb, l, axis=0), np.roll(vb, 3 * l + 1, axis=0)
            self.cache.append((np.ascontiguousarray(k), np.ascontiguousarray(v)))
            for which, a in ((0, self.cache[-1][0]), (1, self.cache[-1][1])):
                M._check(lib.mi355_llama_kv_copy(gm.h, l, which, a.ctypes.data, a.nbytes, 1), "kv_copy")
        self.rng = rng
        self.stream = torch.cuda.Stream()
        self.log(f"mixtral: fp8 KV pool in both models: {time.time() - t0:.1f}s")

    def run(self, ctx=4097, steps=2):
        # the oracle's mat-vecs go through the C twin for the duration of this run only (the module-level hook is restored: other
        # tests in the same process use weight types the C product does not know, e.g. Q8_0 shards)
        keep = OL._qmm
        OL._qmm = self._fast_qmm
        try:
            return self._run(ctx, steps)
        finally:
            OL._qmm = keep

    def _run(self, ctx, steps):
        cfg, gm, rng, M, torch = self.cfg, self.gm, self.rng, self.M, self.torch
        lib = M.lib
        bs, hid, NL = cfg.block_size, cfg.hidden, cfg.n_layers
        blocks = [int(x) for x in (np.arange(-(-(ctx + steps) // bs)) + 1)]
        seqs = [{"tokens": [0] * (ctx - 1) + [int(rng.integers(0, cfg.vocab))], "block_table": blocks}]
        bt = np.zeros((1, self.bps), np.uint32)
        bt[0, : len(blocks)] = blocks
        st = self.stream.cuda_stream
        hip = ctypes.CDLL("libamdhip64.so")
        hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        hip.hipMemcpy.restype = ctypes.c_int
        lib.mi355_llama_act_ptr.restype = ctypes.c_void_p
        xs_ptr = lib.mi355_llama_act_ptr(gm.h, 0)
        # ---- (1) every layer alone, from the oracle's stream (eager launches through mi355_llama_run_part)
        gm.set_graph(False)
        gm.decode_begin([seqs[0]["tokens"][-1]], [ctx], bt, ctx_cap=ctx + steps, stream=st)
        meta = O.prepare_decode(seqs, bs)
        meta["block_tables"] = bt
        cache0 = [(k.copy(), v.copy()) for k, v in self.cache]            # the oracle's step writes the new token's K/V: keep the pool for the replay below
        trace = []
        t0 = time.time()
        ref = self.orc.forward(meta, cache0, trace=trace)
        t_orc = time.time() - t0
        x_in = [np.ascontiguousarray(self.W["tok_embd"][meta["input_ids"]], np.float32)] + trace[:-1]
        got = np.empty((1, hid), np.float32)
        per_layer = []
        for l in range(NL):
            M._check(hip.hipMemcpy(xs_ptr, x_in[l].ctypes.data, x_in[l].nbytes, 1), "H2D")
            for part in range(5):
                M._check(lib.mi355_llama_run_part(gm.h, l, part, st), "run_part")
            torch.cuda.synchronize()
            M._check(hip.hipMemcpy(got.ctypes.data, xs_ptr, got.nbytes, 2), "D2H")
            added = float(np.abs(trace[l] - x_in[l]).max())
            per_layer.append(float(np.abs(got - trace[l]).max() / added))
        M._check(hip.hipMemcpy(xs_ptr, trace[-1].ctypes.data, trace[-1].nbytes, 1), "H2D")
        M._check(lib.mi355_llama_run_part(gm.h, 0, 5, st), "run_part head")
        lg = gm.logits_numpy(1)
        head = float(np.abs(lg - ref).max() / np.abs(ref).max())
        res = {"leg": "mixtral_fp8", "batch": 1, "ctx": ctx, "layers": NL, "worst_layer_rel_err": max(per_layer), "worst_layer": int(np.argmax(per_layer)),
               "median_layer_rel_err": float(np.median(per_layer)), "lm_head_rel_err": head, "oracle": "O1 (unpinned), fp8 KV", "oracle_s_per_step": round(t_orc, 1),
               "units": "layer errors relative to what the layer adds to the stream (attention through the e4m3 cache + the routed experts)"}
        # ---- (2) greedy steps end to end, hipGraph replay; the device pool is restored first (the layer runs above wrote K/V)
        for l in range(NL):
            for which, a in ((0, self.cache[l][0]), (1, self.cache[l][1])):
                M._check(lib.mi355_llama_kv_copy(gm.h, l, which, a.ctypes.data, a.nbytes, 1), "kv_copy")
        gm.set_graph(True)
        gm.decode_begin([seqs[0]["tokens"][-1]], [ctx], bt, ctx_cap=ctx + steps, stream=st)
        cache1 = [(k.copy(), v.copy()) for k, v in self.cache]
        worst, equal, ties, done = 0.0, True, 0, 0
        for step in range(steps):
            gm.decode_step(st)
            tok = int(gm.read_tokens(st)[0])
            g = gm.logits_numpy(1)[0]
            meta = O.prepare_decode(seqs, bs)
            meta["block_tables"] = bt
            r = self.orc.forward(meta, cache1)[0]
            err = float(np.abs(g - r).max())
            worst = max(worst, err / float(np.abs(r).max()))
            done += 1
            want = int(r.argmax())
            if tok != want:
                top2 = np.partition(r, -2)[-2:]
                if float(top2[1] - top2[0]) <= 2.0 * err:
                    ties += 1                                             # near tie: the oracle follows the GPU's token
                    want = tok
                else:
                    equal = False
                    break
            seqs[0]["tokens"].append(want)
        res.update({"steps_compared": done, "logits_max_rel_err": worst, "tokens_equal": bool(equal), "near_tie_tokens": ties})
        return res



    # ------------------------------------------------------------------------------------------------ batch of ragged sequences
    def run_batch(self, seq_lens, steps=1, per_layer=False):
        """the timed batch-32 geometry of bench_legs.py `mixtral_fp8_b32`: ragged contexts, the (token, slot) pairs grouped by expert on the
        device, fp8 KV cache, hipGraph replay -- greedy steps end to end against OracleLlama (O1f products)."""
        keep = OL._qmm
        OL._qmm = lambda x, tw, o2: cref.qmatmul(np.ascontiguousarray(x, np.float32), tw[1], tw[0], 2)
        try:
            cfg, gm, rng = self.cfg, self.gm, self.rng
            bs, B = cfg.block_size, len(seq_lens)
            seqs, nxt = [], 1
            for L in seq_lens:
                n = -(-(int(L) + steps) // bs)
                seqs.append({"tokens": [0] * (int(L) - 1) + [int(rng.integers(0, cfg.vocab))], "block_table": list(range(nxt, nxt + n))})
                nxt += n
            assert nxt <= self.num_blocks, (nxt, self.num_blocks)
            bt = np.zeros((B, self.bps), np.uint32)
            for i, q in enumerate(seqs):
                bt[i, : len(q["block_table"])] = q["block_table"]
            st = self.stream.cuda_stream
            first_ref, layer_res = None, {}
            if per_layer:
                # ---- every layer alone at the batch, teacher-forced from the oracle's stream (eager launches through mi355_llama_run_part:
                # the grouped-expert launches, the fp8 attention stream, the chained images -- what the captured step runs).  The oracle
                # forward that records the stream is also the reference of the first end-to-end step below (same inputs, same pool).
                M, torch = self.M, self.torch
                lib = M.lib
                hip = ctypes.CDLL("libamdhip64.so")
                hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
                hip.hipMemcpy.restype = ctypes.c_int
                lib.mi355_llama_act_ptr.restype = ctypes.c_void_p
                xs_ptr = lib.mi355_llama_act_ptr(gm.h, 0)
                gm.set_graph(False)
                gm.decode_begin([q["tokens"][-1] for q in seqs], [len(q["tokens"]) for q in seqs], bt, ctx_cap=int(max(seq_lens)) + steps, stream=st)
                meta = O.prepare_decode(seqs, bs)
                meta["block_tables"] = bt
                cache0 = [(k.copy(), v.copy()) for k, v in self.cache]
                trace = []
                first_ref = self.orc.forward(meta, cache0, trace=trace)
                x_in = [np.ascontiguousarray(self.W["tok_embd"][meta["input_ids"]], np.float32)] + trace[:-1]
                got = np.empty((B, cfg.hidden), np.float32)
                errs, rows_all = [], []
                for l in range(cfg.n_layers):
                    xi = np.ascontiguousarray(x_in[l], np.float32)
                    M._check(hip.hipMemcpy(xs_ptr, xi.ctypes.data, xi.nbytes, 1), "H2D")
                    # the previous layer's last launch staged THIS layer's q|k|v activation image from its own output (chain hint keyed by the
                    # stream pointer): forget it, or the teacher-forced input just uploaded would never be read (first run of this test:
                    # 3e-2 -- the accumulated distance of the GPU's own stream, not one layer's error)
                    lib.mi355_internal_qmm_set_exact(0)
                    for part in range(5):
                        M._check(lib.mi355_llama_run_part(gm.h, l, part, st), "run_part")
                    torch.cuda.synchronize()
                    M._check(hip.hipMemcpy(got.ctypes.data, xs_ptr, got.nbytes, 2), "D2H")
                    added = np.abs(trace[l] - xi).max(axis=1)
                    row_err = np.abs(got - trace[l]).max(axis=1) / added
                    rows_all.append(row_err)
                    errs.append(float(row_err.max()))
                rows_all = np.asarray(rows_all)                             # [layer][row]
                layer_res = {"worst_layer_rel_err": max(errs), "worst_layer": int(np.argmax(errs)), "median_layer_rel_err": float(np.median(errs)),
                             "median_over_all_rows_and_layers": float(np.median(rows_all)), "p90_over_all_rows_and_layers": float(np.percentile(rows_all, 90)),
                             "rows_of_worst_layer": [round(float(x), 5) for x in rows_all[int(np.argmax(errs))]],
                             "per_layer": [round(e, 5) for e in errs], "ctx_of_rows": [int(x) for x in seq_lens]}
                for l in range(cfg.n_layers):                              # the layer runs wrote K/V of the new token: restore the device pool
                    for which, a in ((0, self.cache[l][0]), (1, self.cache[l][1])):
                        M._check(lib.mi355_llama_kv_copy(gm.h, l, which, a.ctypes.data, a.nbytes, 1), "kv_copy")
            gm.set_graph(True)
            gm.decode_begin([q["tokens"][-1] for q in seqs], [len(q["tokens"]) for q in seqs], bt, ctx_cap=int(max(seq_lens)) + steps, stream=st)
            # with per_layer the traced oracle forward above WAS step 0 (same inputs, same pool; it left the new token's K/V in cache0)
            cache1 = cache0 if first_ref is not None else [(k.copy(), v.copy()) for k, v in self.cache]
            worst, equal, ties, done, t_orc = 0.0, True, 0, 0, 0.0
            for step in range(steps):
                gm.decode_step(st)
                toks = [int(t) for t in gm.read_tokens(st)]
                got = gm.logits_numpy(B)
                meta = O.prepare_decode(seqs, bs)
                meta["block_tables"] = bt
                t0 = time.time()
                ref = first_ref if (step == 0 and first_ref is not None) else self.orc.forward(meta, cache1)
                t_orc += time.time() - t0
                done += 1
                for b in range(B):
                    err = float(np.abs(got[b] - ref[b]).max())
                    worst = max(worst, err / float(np.abs(ref[b]).max()))
                    want = int(ref[b].argmax())
                    if toks[b] != want:
                        top2 = np.partition(ref[b], -2)[-2:]
                        if float(top2[1] - top2[0]) <= 2.0 * err:
                            ties += 1
                            want = toks[b]
                        else:
                            equal = False
                    seqs[b]["tokens"].append(want)
                if not equal:
                    break
            layer_res.update({"leg": "mixtral_fp8_b32", "batch": B, "ctx_max": int(max(seq_lens)), "layers": cfg.n_layers, "steps_compared": done,
                    "logits_max_rel_err": worst, "tokens_equal": bool(equal), "near_tie_tokens": ties,
                    "oracle": "O1f (unpinned), fp8 KV, experts per token", "oracle_s_per_step": round(t_orc / max(done, 1), 1)})
            return layer_res
        finally:
            OL._qmm = keep

    # ------------------------------------------------------------------------------------------------ configs[4]: chunked prefill at size
    def run_chunked_prompt(self, T=16384, chunk=8192):
        """BASELINE configs[4]'s prompt: T tokens prefilled in chunks of `chunk` through the C++ scheduler (prefill_chunk_size,
        pipelines/inputs.rs:90-230): every chunk after the first attends over its predecessors through the e4m3 cache (cached prefix).
        For a ONE-layer model of full width the oracle needs only cheap pieces: q / k / v of all T tokens (one O1f product each), the
        e4m3 cache bytes, and -- for the LAST token, whose logits the second chunk returns -- attention over all T keys, the routed
        experts, the head.  Compared: the device cache bytes of the whole prompt (both chunks' writes) and the final logits."""
        from candle_vllm_amd import block_engine as be
        from oracle import llama as OL2
        cfg, gm, M, rng = self.cfg, self.gm, self.M, self.rng
        assert cfg.n_layers == 1, "the oracle restatement below is written for one layer"
        bs, hid, H, Hkv, D = cfg.block_size, cfg.hidden, cfg.n_heads, cfg.n_kv_heads, cfg.head_dim
        prompt = [int(t) for t in rng.integers(0, cfg.vocab, T)]
        sched = be.Scheduler(block_size=bs, num_gpu_blocks=self.num_blocks, num_cpu_blocks=0, max_num_parallel_reqs=1,
                             max_num_batched_tokens=chunk, prefill_chunk_size=chunk)
        eng = sched.block_engine
        seq = eng.new_sequence(0, prompt)
        sched.add_sequence(0, [seq])
        chunks, got, metas = [], None, []
        while True:
            out = sched.schedule()
            assert out.is_prompt and out.scheduled == [0]
            meta = eng.prepare_prompt([seq], chunk=chunk)
            metas.append(meta)
            lg = gm.forward_prefill(meta)
            chunks.append(len(meta["input_ids"]))
            if sched.filter_prefill_finished(out.scheduled):
                got = lg.cpu().numpy()[0]
                break
        assert sum(chunks) == T and len(chunks) == -(-T // chunk), chunks
        assert int(metas[-1]["context_lens"][0]) == T                  # the last chunk saw the whole prompt as its context
        kc_dev, vc_dev = gm.kv_download_u8(0)
        # ---- oracle, layer 0
        t0 = time.time()
        lw, W = self.W["layers"][0], self.W
        qmm = lambda x, tw: cref.qmatmul(np.ascontiguousarray(x, np.float32), tw[1], tw[0], 2)      # O1f (f32 blocked dots)
        xs = W["tok_embd"][np.asarray(prompt)].astype(np.float32)
        x = O.rms_norm(xs, lw["attn_norm"], cfg.rms_eps)
        cos, sin = O.rope_tables(cfg.rope_theta, D, cfg.max_seq)
        pos = np.arange(T)
        k = O.rope_apply(qmm(x, lw["wk"]).reshape(T, Hkv, D), cos, sin, pos, interleaved=True)
        v = qmm(x, lw["wv"]).reshape(T, Hkv, D)
        table = np.concatenate([m["block_tables"][0] for m in metas[-1:]])          # the full table of the sequence
        slots = np.concatenate([m["slot_mapping"] for m in metas])
        ks, vs = O.kv_cache_shapes(self.num_blocks, bs, Hkv, D, 1, False)
        kc, vc = np.zeros(ks, np.uint8), np.zeros(vs, np.uint8)
        O.reshape_and_cache_fp8(O.round_bf16(k), O.round_bf16(v), kc, vc, slots, False)
        used = np.unique(table[: -(-T // bs)])
        kq_eq = float((kc_dev[used] == kc[used]).mean())
        vq_eq = float((vc_dev[used] == vc[used]).mean())
        # an e4m3 byte differs where the prompt GEMM's 3e-4 moved a bf16 value across an fp8 rounding boundary: never by more than one code
        kd = np.abs(O.e4m3fn_to_f32(kc_dev[used]) - O.e4m3fn_to_f32(kc[used]))
        k_ulp = float((kd / np.maximum(np.abs(O.e4m3fn_to_f32(kc[used])), 2.0 ** -6)).max())
        # the last token: attention over all T keys read back from the ORACLE's e4m3 cache
        q_last = O.rope_apply(qmm(x[-1:], lw["wq"]).reshape(1, H, D), cos, sin, pos[-1:], interleaved=True)
        kb, vb = O.fp8_cache_as_bf16_bits(kc, vc, False)
        kk, vv = O.gather_kv(kb, vb, table, T, False)
        y = O.prefill_attention(O.round_bf16(q_last), O.bf16_bits_to_f32(kk), O.bf16_bits_to_f32(vv), 1.0 / np.sqrt(float(D)), cached=T - 1)
        x1 = qmm(y.reshape(1, H * D), lw["wo"]) + xs[-1:]
        keep = OL2._qmm
        OL2._qmm = lambda a, tw, o2: qmm(a, tw)
        try:
            x2 = OL2.moe_forward(O.rms_norm(x1, lw["ffn_norm"], cfg.rms_eps), lw, 2) + x1
        finally:
            OL2._qmm = keep
        ref = qmm(O.rms_norm(x2, W["output_norm"], cfg.rms_eps), W["output"])[0]
        err = float(np.abs(got - ref).max() / np.abs(ref).max())
        return {"leg": "mixtral_chunked_prefill", "tokens": T, "chunks": chunks, "layers": 1, "logits_max_rel_err": err,
                "tokens_equal": bool(int(got.argmax()) == int(ref.argmax())), "k_cache_bytes_equal": kq_eq, "v_cache_bytes_equal": vq_eq,
                "k_cache_max_rel_diff": k_ulp, "oracle": "O1f (unpinned), fp8 KV", "oracle_s": round(time.time() - t0, 1)}
