"""Full-size parity leg (TEST INFRASTRUCTURE: used by tests/test_gpu_fullsize.py and bench.py's `parity` object).

The geometry bench.py times -- Llama-3-8B shapes, Q4_K_M mixture, 4096-token context in the paged KV cache, block 64 --
built ONCE from native GGUF blocks in the C oracle (oracle/oracle.c) and handed byte for byte to the GPU model
(`mi355_llama_set_qweight`), then the same step run on both:
  * decode, batch 1, hipGraph replay          (the headline launches: qmm_kernel wgs = 896 / 2004, MFMA attention at ctx 4.1k)
  * decode, batch 32, ragged contexts         (the chained wide path at hidden 4096)
  * one prompt step of T tokens               (the prompt-step GEMM path + K4 prefill attention)
Reference order of operations: src/openai/models/quantized_llama.rs:424-506, layers/attention.rs:910-1011.
Oracle arithmetic: O1 (dequantise, f64 dot) for batch 1; O1f (same definition, f32 blocked dots -- ~1e-6 from O1,
tests/test_cpu_oracle.py) for the many-token steps so they finish in seconds.  PARITY UNPINNED (no reference fixture
exists for the float kernels): this leg checks the GPU path against the restatement at full size.
"""
import ctypes
import time

import numpy as np

from oracle import cref
from oracle import ops as O
from oracle.llama import LlamaConfig, q4km_type_for

_NAMES = ["wq", "wk", "wv", "wo", "w1", "w2", "w3"]
_BLOCK_BYTES = {12: 144, 14: 210}


def flash_to_paged(k, v):
    """[NB,bs,Hkv,D] -> K [NB,Hkv,D/8,bs,8], V [NB,Hkv,D,bs]   (cache_engine.rs:298-341)"""
    NB, bs, Hkv, D = k.shape
    kp = np.ascontiguousarray(k.reshape(NB, bs, Hkv, D // 8, 8).transpose(0, 2, 3, 1, 4))
    vp = np.ascontiguousarray(v.transpose(0, 2, 3, 1))
    return kp, vp


def paged_to_flash(kp, vp):
    NB, Hkv, D8, bs, x = kp.shape
    k = np.ascontiguousarray(kp.transpose(0, 3, 1, 2, 4).reshape(NB, bs, Hkv, D8 * x))
    v = np.ascontiguousarray(vp.transpose(0, 3, 1, 2))
    return k, v


class Pair:
    """the C oracle model and the GPU model over the same weight bytes, plus a shared random KV block pool"""

    def __init__(self, cfg=None, seed=1235, num_blocks=448, max_batch=32, log=None, fill_scale=1.0):
        """fill_scale: std of the synthetic weights relative to the bench's (1.0 = std ~0.04: every branch has gain >> 1 and
        the 32-layer stack amplifies rounding noise chaotically, fine for the per-layer checks; 0.2 = branch gain < 1 as in
        a trained checkpoint, for the end-to-end logits checks)"""
        import torch
        from candle_vllm_amd import model as M
        self.M, self.torch = M, torch
        self.cfg = cfg or LlamaConfig.llama3_8b()
        cfg = self.cfg
        self.log = log or (lambda *a: None)
        t0 = time.time()
        cref.build()
        types = [q4km_type_for(n, l, cfg.n_layers) for l in range(cfg.n_layers) for n in _NAMES]
        types.append(q4km_type_for("output", 0, cfg.n_layers))
        self.orc = cref.CLlama(cfg, W=None, types=types, seed=seed, fill_scale=fill_scale)
        self.fill_scale = fill_scale
        rng = np.random.default_rng(seed)
        base = (rng.standard_normal((1002, cfg.hidden)) * 0.02).astype(np.float32)
        emb = np.ascontiguousarray(np.tile(base, (-(-cfg.vocab // 1002), 1))[: cfg.vocab])
        self.max_blocks = -(-(4096 + 64) // cfg.block_size) + 1
        self.gm = M.GGUFLLaMa(cfg, max_batch=max_batch, max_blocks_per_seq=self.max_blocks, kv_layout=M.KV_PAGED)
        gm, lib = self.gm, M.lib

        def f32(layer, o_which, g_which, a):
            a = np.ascontiguousarray(a, np.float32)
            self.orc.set_f32(layer, o_which, a)
            M._check(lib.mi355_llama_set_f32(gm.h, layer, g_which, a.ctypes.data, a.size), "set_f32")
        f32(-1, 9, M.W_TOK_EMBD, emb)
        f32(-1, 10, M.W_OUTPUT_NORM, 1.0 + rng.normal(0, 0.02, cfg.hidden))
        for l in range(cfg.n_layers):
            f32(l, 7, M.W_ATTN_NORM, 1.0 + rng.normal(0, 0.02, cfg.hidden))
            f32(l, 8, M.W_FFN_NORM, 1.0 + rng.normal(0, 0.02, cfg.hidden))
        for l in list(range(cfg.n_layers)) + [-1]:
            for w in (range(7) if l >= 0 else [11]):
                p, t, n, k = self.orc.qweight(l, w)
                M._check(lib.mi355_llama_set_qweight(gm.h, l, M.W_OUTPUT if l < 0 else w, t, p, n, k), "set_qweight")
                gm.weight_bytes += n * (k // 256) * _BLOCK_BYTES[t]
        self.log(f"weights in both models: {time.time() - t0:.1f}s ({gm.weight_bytes / 1e9:.2f} GB)")
        # ---- KV block pool: random bf16 K/V in every block (flash layout on the oracle side), one base pool rolled per layer
        t0 = time.time()
        self.num_blocks = num_blocks
        gm.alloc_kv_cache(num_blocks)
        shape = (num_blocks, cfg.block_size, cfg.n_kv_heads, cfg.head_dim)
        kb = (rng.standard_normal(shape, dtype=np.float32).view(np.uint32) >> 16).astype(np.uint16)
        vb = (rng.standard_normal(shape, dtype=np.float32).view(np.uint32) >> 16).astype(np.uint16)
        self.cache = []
        for l in range(cfg.n_layers):
            k, v = np.roll(kb, l, axis=0), np.roll(vb, 3 * l + 1, axis=0)
            self.cache.append((k, v))
            gm.kv_upload(l, *flash_to_paged(k, v))
        self.perm = rng.permutation(num_blocks - 1) + 1            # shuffled physical blocks (block 0 unused)
        self._next = 0
        self.rng = rng
        self.stream = torch.cuda.Stream()
        self.log(f"KV pool ({num_blocks} blocks) in both models: {time.time() - t0:.1f}s")

    def take_blocks(self, n):
        if self._next + n > len(self.perm):
            raise RuntimeError("block pool exhausted")
        b = self.perm[self._next: self._next + n]
        self._next += n
        return [int(x) for x in b]

    # ------------------------------------------------------------------------------------------------ decode
    def run_decode(self, seq_lens, steps=2, o2=0, graph=True):
        """`steps` greedy decode steps of len(seq_lens) sequences whose first seq_len-1 tokens are already in the cache.
        Returns a dict: max_rel_err (max over steps and sequences of max|got-ref| / max|ref| of that row), tokens_equal (no
        mismatch outside a near tie), near_tie_tokens (mismatches where the oracle's own top-2 gap is below the error bound: the
        oracle then follows the GPU's token and the comparison continues), steps_compared."""
        cfg, gm, rng = self.cfg, self.gm, self.rng
        B, bs = len(seq_lens), cfg.block_size
        self._next = 0                                                 # scenarios reuse the pool (both caches hold the same bytes)
        seqs = []
        for L in seq_lens:
            blocks = self.take_blocks(-(-(int(L) + steps) // bs))
            seqs.append({"tokens": [0] * (int(L) - 1) + [int(rng.integers(0, cfg.vocab))], "block_table": blocks})
        bt = np.zeros((B, self.max_blocks), np.uint32)
        for i, s in enumerate(seqs):
            bt[i, : len(s["block_table"])] = s["block_table"]
        st = self.stream.cuda_stream
        gm.set_graph(bool(graph))
        gm.decode_begin([s["tokens"][-1] for s in seqs], [len(s["tokens"]) for s in seqs], bt,
                        ctx_cap=int(max(seq_lens)) + steps, stream=st)
        worst, worst_b, equal, ties, t_orc, done = 0.0, 0.0, True, 0, 0.0, 0
        for step in range(steps):
            gm.decode_step(st)
            got_tok = [int(t) for t in gm.read_tokens(st)]
            got = gm.logits_numpy(B)
            meta = O.prepare_decode(seqs, bs)
            meta["block_tables"] = bt                                  # same padded width as the GPU step
            if step == 0:
                # The reference's CPU path runs attention on bf16 TENSORS (models/mod.rs:1288-1306): its score matmul, its
                # softmax and its P.V matmul each round to bf16, the oracle's parity target keeps them in f32.  How far do
                # those rounding points alone move the logits through the 32-layer stack?  That is the floor under any
                # end-to-end comparison (the GPU kernel rounds P to bf16 for the MFMA, like the reference, at other points).
                cref.lib().orc_llama_set_attn_bf16(1)
                other = self.orc.decode(meta, self.cache, o2=o2)
                cref.lib().orc_llama_set_attn_bf16(0)
            t0 = time.time()
            ref = self.orc.decode(meta, self.cache, o2=o2)
            t_orc += time.time() - t0
            if step == 0:
                spread = float((np.abs(other - ref).max(axis=1) / np.abs(ref).max(axis=1)).max())
                # the GPU against the bf16-attention oracle itself (the faithful restatement of the reference's CPU path)
                worst_b = float((np.abs(got - other).max(axis=1) / np.abs(other).max(axis=1)).max())
            done += 1
            for b in range(B):
                scale = float(np.abs(ref[b]).max())
                err = float(np.abs(got[b] - ref[b]).max())
                worst = max(worst, err / scale)
                want = int(ref[b].argmax())
                if got_tok[b] != want:
                    top2 = np.partition(ref[b], -2)[-2:]
                    if float(top2[1] - top2[0]) <= 2.0 * err:
                        # a near tie (the oracle's own top-2 gap is inside the error bound) is not a mismatch; the device loop
                        # has fed ITS token, so the oracle follows it for this sequence and the comparison goes on
                        ties += 1
                        want = got_tok[b]
                    else:
                        equal = False
                seqs[b]["tokens"].append(want)
            if not equal:
                break                                                  # a real mismatch: the two runs have diverged
        return {"batch": B, "steps": step + 1, "steps_compared": done, "ctx_max": int(max(seq_lens)), "graph": bool(graph),
                "max_rel_err": worst, "max_rel_err_vs_bf16_attention": worst_b, "tokens_equal": bool(equal), "near_tie_tokens": ties,
                "reference_bf16_attention_spread": spread,
                "oracle": "O1 (unpinned)" if int(o2) == 0 else "O1f (unpinned)", "oracle_s_per_step": round(t_orc / (step + 1), 2)}

    def run_decode_faithful(self, seq_lens, steps=2, o2=0, graph=True, exact=False):
        """north_star's comparison itself: "within 1e-3 of the reference CPU logits".  The reference's CPU path rounds the attention
        scores, the scaled scores, the probabilities and P.V to bf16 (NaiveAttention::forward on bf16 tensors, models/mod.rs:1288-1306);
        through 32 layers those points alone move the logits by 1.7-3 % (run_decode's `reference_bf16_attention_spread`), so the
        product attention kernels (f32 scores and probabilities) cannot be within 1e-3 of it and neither can anything else that
        rounds elsewhere.  Here BOTH sides take the reference's points: the GPU model in parity mode
        (mi355_llama_set_attention_numerics(1): mi355_paged_attention_reference_numerics, same summation orders as the oracle) against
        oracle.c's bf16-attention mode -- every other kernel of the step (quantised mat-muls, RMSNorm, RoPE, cache write, residuals,
        lm_head, argmax) is the product's.  Returns the worst logit error relative to the row's largest logit.
        exact=True (parity mode 2, csrc/qmm_exact.inc): the step's mat-vecs exact to f32 rounding as well -- the run that shows
        whether the product's distance is amplified per-product rounding noise (it then falls to the oracle's self-spread) or a bias."""
        cfg, gm, rng = self.cfg, self.gm, self.rng
        B, bs = len(seq_lens), cfg.block_size
        self._next = 0
        seqs = []
        for L in seq_lens:
            blocks = self.take_blocks(-(-(int(L) + steps) // bs))
            seqs.append({"tokens": [0] * (int(L) - 1) + [int(rng.integers(0, cfg.vocab))], "block_table": blocks})
        bt = np.zeros((B, self.max_blocks), np.uint32)
        for i, s in enumerate(seqs):
            bt[i, : len(s["block_table"])] = s["block_table"]
        st = self.stream.cuda_stream
        gm.set_attention_numerics(2 if exact else 1)
        cref.lib().orc_llama_set_attn_bf16(1)
        try:
            gm.set_graph(bool(graph))
            gm.decode_begin([s["tokens"][-1] for s in seqs], [len(s["tokens"]) for s in seqs], bt,
                            ctx_cap=int(max(seq_lens)) + steps, stream=st)
            worst, equal, ties, done = 0.0, True, 0, 0
            per_step, self_spread = [], None
            for step in range(steps):
                gm.decode_step(st)
                got_tok = [int(t) for t in gm.read_tokens(st)]
                got = gm.logits_numpy(B)
                meta = O.prepare_decode(seqs, bs)
                meta["block_tables"] = bt
                if step == 0 and B == 1:
                    # How reproducible is the reference's arithmetic itself?  The SAME restatement with its mat-vecs summed in blocked f32
                    # (O1f) instead of f64 (O1) -- ~1e-6 apart per product, what two thread counts of a CPU backend differ by -- run through
                    # the same 32 layers with the same bf16 attention tensors: the distance of the two logit rows is the floor under ANY
                    # end-to-end comparison with the reference CPU path (tools/exp_oracle_self_spread.py: 3.3e-3 at this geometry).
                    alt = self.orc.decode(meta, self.cache, o2=2 if int(o2) == 0 else 0)
                ref = self.orc.decode(meta, self.cache, o2=o2)
                if step == 0 and B == 1:
                    self_spread = float((np.abs(alt - ref).max(axis=1) / np.abs(ref).max(axis=1)).max())
                done += 1
                w = 0.0
                for b in range(B):
                    err = float(np.abs(got[b] - ref[b]).max())
                    w = max(w, err / float(np.abs(ref[b]).max()))
                    want = int(ref[b].argmax())
                    if got_tok[b] != want:
                        top2 = np.partition(ref[b], -2)[-2:]
                        if float(top2[1] - top2[0]) <= 2.0 * err:
                            ties += 1
                            want = got_tok[b]
                        else:
                            equal = False
                    seqs[b]["tokens"].append(want)
                per_step.append(w)
                worst = max(worst, w)
                if not equal:
                    break
        finally:
            gm.set_attention_numerics(0)
            cref.lib().orc_llama_set_attn_bf16(0)
        return {"batch": B, "steps_compared": done, "ctx_max": int(max(seq_lens)), "graph": bool(graph), "max_rel_err": worst,
                "per_step": [round(x, 7) for x in per_step], "tokens_equal": bool(equal), "near_tie_tokens": ties,
                "oracle_self_spread_f64_vs_f32_dots": self_spread,
                "mode": "reference-faithful attention numerics on both sides (models/mod.rs:1288-1306)" + (" + exact mat-vecs (qmm_exact.inc)" if exact else ""),
                "oracle": "O1 + bf16 attention tensors (unpinned)" if int(o2) == 0 else "O1f + bf16 attention tensors (unpinned)"}

    # ------------------------------------------------------------------------------------------------ one layer at a time
    def run_layerwise(self, seq_lens, o2=0):
        """Teacher-forced: the oracle's decode step records the residual stream at every layer entry; the GPU runs each
        layer's five launch groups (mi355_llama_run_part) from the ORACLE's layer input and is compared with the oracle's
        layer output -- no error amplification through the stack, every launch at its full-size geometry.  Returns the
        worst per-layer error relative to what the layer ADDED to the stream, and the lm_head error from the oracle's
        final hidden state."""
        cfg, gm, rng, M = self.cfg, self.gm, self.rng, self.M
        lib = M.lib
        B, bs, hid, NL = len(seq_lens), cfg.block_size, cfg.hidden, cfg.n_layers
        self._next = 0
        seqs = []
        for L in seq_lens:
            blocks = self.take_blocks(-(-(int(L) + 1) // bs))
            seqs.append({"tokens": [0] * (int(L) - 1) + [int(rng.integers(0, cfg.vocab))], "block_table": blocks})
        bt = np.zeros((B, self.max_blocks), np.uint32)
        for i, s in enumerate(seqs):
            bt[i, : len(s["block_table"])] = s["block_table"]
        st = self.stream.cuda_stream
        gm.set_graph(False)
        gm.decode_begin([s["tokens"][-1] for s in seqs], [len(s["tokens"]) for s in seqs], bt,
                        ctx_cap=int(max(seq_lens)) + 1, stream=st)
        meta = O.prepare_decode(seqs, bs)
        meta["block_tables"] = bt
        trace = np.zeros((NL + 1, B, hid), np.float32)
        self.orc.set_trace(trace)
        ref = self.orc.decode(meta, self.cache, o2=o2)
        self.orc.set_trace(None)
        hip = ctypes.CDLL("libamdhip64.so")
        hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        hip.hipMemcpy.restype = ctypes.c_int
        xs = lib.mi355_llama_act_ptr(gm.h, 0)
        got = np.empty((B, hid), np.float32)
        per_layer = []
        for l in range(NL):
            x_in = np.ascontiguousarray(trace[l])
            M._check(hip.hipMemcpy(xs, x_in.ctypes.data, x_in.nbytes, 1), "hipMemcpy H2D")
            lib.mi355_internal_qmm_set_exact(lib.mi355_internal_qmm_get_exact())    # forget the chain hint: the previous layer's epilogue staged this layer's q|k|v image from ITS output, not from the teacher-forced input
            for part in range(5):
                M._check(lib.mi355_llama_run_part(gm.h, l, part, st), "run_part")
            self.torch.cuda.synchronize()
            M._check(hip.hipMemcpy(got.ctypes.data, xs, got.nbytes, 2), "hipMemcpy D2H")
            added = np.abs(trace[l + 1] - trace[l]).max(axis=1)            # per sequence: what the layer added
            err = np.abs(got - trace[l + 1]).max(axis=1)
            per_layer.append(float((err / added).max()))
        x_in = np.ascontiguousarray(trace[NL])
        M._check(hip.hipMemcpy(xs, x_in.ctypes.data, x_in.nbytes, 1), "hipMemcpy H2D")
        M._check(lib.mi355_llama_run_part(gm.h, 0, 5, st), "run_part head")
        lg = gm.logits_numpy(B)
        head = float((np.abs(lg - ref).max(axis=1) / np.abs(ref).max(axis=1)).max())
        tok_ok = [int(r.argmax()) for r in lg] == [int(r.argmax()) for r in ref]
        return {"batch": B, "ctx_max": int(max(seq_lens)), "worst_layer_rel_err": max(per_layer),
                "worst_layer": int(np.argmax(per_layer)), "lm_head_rel_err": head, "tokens_equal": tok_ok,
                "per_layer": [round(e, 6) for e in per_layer],
                "oracle": "O1 (unpinned)" if int(o2) == 0 else "O1f (unpinned)"}

    # ------------------------------------------------------------------------------------------------ one launch group at a time
    def run_parts(self, seq_lens, o2=0, kv_layers=(0, 15, 31)):
        """Every launch group of every layer from the ORACLE's own inputs (same bytes in, so no rounding-point noise is
        carried from one group into the next): QKV -> q and the K/V it writes (rounding points: <= 1 bf16 ulp), attention
        (<= 1 bf16 ulp of the largest output, the bound of test_gpu_ops.py), wo / gate-up / down (f32: relative to what the
        group produces).  Chaining is off so each launch stages the activations that were uploaded for it."""
        cfg, gm, rng, M = self.cfg, self.gm, self.rng, self.M
        lib = M.lib
        B, bs, hid, NL, I = len(seq_lens), cfg.block_size, cfg.hidden, cfg.n_layers, cfg.intermediate
        HD = cfg.n_heads * cfg.head_dim
        self._next = 0
        seqs = []
        for L in seq_lens:
            blocks = self.take_blocks(-(-(int(L) + 1) // bs))
            seqs.append({"tokens": [0] * (int(L) - 1) + [int(rng.integers(0, cfg.vocab))], "block_table": blocks})
        bt = np.zeros((B, self.max_blocks), np.uint32)
        for i, s in enumerate(seqs):
            bt[i, : len(s["block_table"])] = s["block_table"]
        st = self.stream.cuda_stream
        gm.set_graph(False)
        gm.decode_begin([s["tokens"][-1] for s in seqs], [len(s["tokens"]) for s in seqs], bt,
                        ctx_cap=int(max(seq_lens)) + 1, stream=st)
        meta = O.prepare_decode(seqs, bs)
        meta["block_tables"] = bt
        tr_xs = np.zeros((NL + 1, B, hid), np.float32)
        tr_q, tr_att = np.zeros((NL, B, HD), np.float32), np.zeros((NL, B, HD), np.float32)
        tr_mid, tr_h = np.zeros((NL, B, hid), np.float32), np.zeros((NL, B, I), np.float32)
        self.orc.set_trace(tr_xs)
        self.orc.set_trace_parts(tr_q, tr_att, tr_mid, tr_h)
        self.orc.decode(meta, self.cache, o2=o2)
        self.orc.set_trace(None)
        self.orc.set_trace_parts()
        hip = ctypes.CDLL("libamdhip64.so")
        hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        hip.hipMemcpy.restype = ctypes.c_int
        p_xs, p_q, p_att, p_h = [lib.mi355_llama_act_ptr(gm.h, i) for i in range(4)]

        def up(ptr, a):
            a = np.ascontiguousarray(a)
            M._check(hip.hipMemcpy(ptr, a.ctypes.data, a.nbytes, 1), "hipMemcpy H2D")

        def down(ptr, shape, dtype):
            self.torch.cuda.synchronize()
            a = np.empty(shape, dtype)
            M._check(hip.hipMemcpy(a.ctypes.data, ptr, a.nbytes, 2), "hipMemcpy D2H")
            return a

        def run(l, part):
            M._check(lib.mi355_llama_run_part(gm.h, l, part, st), "run_part")

        def ulp_err(got, ref):      # rounding point after a mat-vec: error beyond ONE bf16 ulp of the element, relative to the row's
            # largest value (the mat-vec itself is good to 1e-4 of that scale, test_gpu_ops.py)
            ref2, got2 = ref.reshape(B, -1), got.reshape(B, -1)
            excess = np.maximum(np.abs(got2 - ref2) - 2.0 ** -7 * np.abs(ref2), 0.0)
            return float((excess.max(axis=1) / np.abs(ref2).max(axis=1)).max())
        slots = np.asarray(meta["slot_mapping"], np.int64)
        worst = {"q_excess": 0.0, "kv_excess": 0.0, "q_flip_frac": 0.0, "attn": 0.0, "wo": 0.0, "gate_up": 0.0, "down": 0.0}
        from candle_vllm_amd import tuning
        with tuning(9, 0):
            for l in range(NL):
                up(p_xs, tr_xs[l]); run(l, 0)
                q_got = O.bf16_bits_to_f32(down(p_q, (B, HD), np.uint16))
                worst["q_excess"] = max(worst["q_excess"], ulp_err(q_got, tr_q[l]))
                worst["q_flip_frac"] = max(worst["q_flip_frac"], float((q_got != tr_q[l]).mean()))
                if l in kv_layers:
                    gk, gv = paged_to_flash(*gm.kv_download(l))
                    for g, o in ((gk, self.cache[l][0]), (gv, self.cache[l][1])):
                        a = O.bf16_bits_to_f32(g.reshape(-1, cfg.n_kv_heads * cfg.head_dim)[slots])
                        b = O.bf16_bits_to_f32(o.reshape(-1, cfg.n_kv_heads * cfg.head_dim)[slots])
                        worst["kv_excess"] = max(worst["kv_excess"], ulp_err(a, b))
                gm.kv_upload(l, *flash_to_paged(*self.cache[l]))            # attention sees exactly the oracle's K/V
                up(p_q, O.f32_to_bf16_bits(tr_q[l])); run(l, 1)
                a_got = O.bf16_bits_to_f32(down(p_att, (B, HD), np.uint16))
                for b in range(B):
                    sc = float(np.abs(tr_att[l, b]).max())
                    worst["attn"] = max(worst["attn"], float(np.abs(a_got[b] - tr_att[l, b]).max() / (2.0 ** -7 * sc + 1e-6)))
                up(p_att, O.f32_to_bf16_bits(tr_att[l])); up(p_xs, tr_xs[l]); run(l, 2)
                mid = down(p_xs, (B, hid), np.float32)
                add = np.abs(tr_mid[l] - tr_xs[l]).max(axis=1)
                worst["wo"] = max(worst["wo"], float((np.abs(mid - tr_mid[l]).max(axis=1) / add).max()))
                up(p_xs, tr_mid[l]); run(l, 3)
                h = down(p_h, (B, I), np.float32)
                worst["gate_up"] = max(worst["gate_up"], float((np.abs(h - tr_h[l]).max(axis=1) / np.abs(tr_h[l]).max(axis=1)).max()))
                up(p_h, tr_h[l]); up(p_xs, tr_mid[l]); run(l, 4)
                out = down(p_xs, (B, hid), np.float32)
                add = np.abs(tr_xs[l + 1] - tr_mid[l]).max(axis=1)
                worst["down"] = max(worst["down"], float((np.abs(out - tr_xs[l + 1]).max(axis=1) / add).max()))
        worst.update({"batch": B, "ctx_max": int(max(seq_lens)), "oracle": "O1 (unpinned)" if int(o2) == 0 else "O1f (unpinned)",
                      "units": "q/kv_excess: error beyond one bf16 ulp of the element, relative to the row's largest value; attn: bf16 "
                               "ulps of the row's largest output; wo/gate_up/down: relative to what the group adds / produces"})
        return worst

    # ------------------------------------------------------------------------------------------------ prompt step
    def run_prompt(self, T=2048, check_layers=(0, 15, 31)):
        """one prompt step of a T-token sequence (no cached prefix): last-token logits + the K/V both sides wrote"""
        cfg, gm, rng = self.cfg, self.gm, self.rng
        bs = cfg.block_size
        self._next = 0
        blocks = self.take_blocks(-(-T // bs))
        seqs = [{"tokens": [int(t) for t in rng.integers(0, cfg.vocab, T)], "block_table": blocks}]
        meta = O.prepare_prompt(seqs, bs)
        with self.torch.cuda.stream(self.stream):
            got = gm.forward_prefill(meta).cpu().numpy()[0]
        t0 = time.time()
        ref = self.orc.prefill(meta["input_ids"], meta["positions"], meta["slot_mapping"], self.cache)
        t_orc = time.time() - t0
        err = float(np.abs(got - ref).max() / np.abs(ref).max())
        top2 = np.partition(ref, -2)[-2:]
        tok_ok = int(got.argmax()) == int(ref.argmax())
        tie = (not tok_ok) and float(top2[1] - top2[0]) <= 2.0 * float(np.abs(got - ref).max())
        kv_err = 0.0
        for l in check_layers:
            if l >= cfg.n_layers:
                continue
            gk, gv = paged_to_flash(*gm.kv_download(l))
            for g, o in ((gk, self.cache[l][0]), (gv, self.cache[l][1])):
                a, b = O.bf16_bits_to_f32(g[blocks]), O.bf16_bits_to_f32(o[blocks])
                kv_err = max(kv_err, float(np.abs(a - b).max() / np.abs(b).max()))
        return {"tokens": T, "max_rel_err": err, "tokens_equal": tok_ok or tie, "near_tie": tie, "kv_max_rel_err": kv_err,
                "oracle": "O1f (unpinned)", "oracle_s": round(t_orc, 1)}


def ragged_batch32(rng):
    """32 ragged contexts that include the benchmarked 4 k context: a few long sequences + many short ones"""
    lens = [4097, 3001, 2049, 1025] + [int(x) for x in rng.integers(64, 257, 28)]
    return lens
