"""Full-size parity legs of the 16-bit host layer (TEST INFRASTRUCTURE: tests/test_gpu_fullsize.py and bench_legs.py's `parity`).

The two geometries bench_legs.py times on `dense_model.cpp`, at their benchmarked sizes, against the bf16-rounding oracle
(oracle/dense_llama.py: every op result rounded to the model dtype, as candle's 16-bit CPU tensors are):
  * bf16_b32   BASELINE configs[2]: Llama-3-8B shapes, 32 layers, batch 32 with ragged contexts up to 4097 tokens in the paged
               cache (llama.rs:139-201, attention.rs:585-734, mlp.rs:440-458, linear.rs:124-172)
  * gptq_qwen2 BASELINE configs[3] on one GPU: Qwen2-7B shapes, 28 layers, every projection GPTQ 4-bit group 128 through the
               marlin_4bit arm, qkv bias, batch 1 at context 4097 (qwen.rs:78-96, gptq.rs:26-204, linear.rs:845-906)
Weights: one random tensor per projection SHAPE, shared by the layers (as bench_legs.py builds its GPTQ leg) -- every launch has
its full-size geometry and the stack its full depth, while the oracle keeps seven f64 matrices instead of 8 G weights; norms and
biases differ per layer.  The same bits go to both sides.  Two checks from ONE oracle forward that records the residual stream
at every layer entry:
  * every layer teacher-forced: the GPU runs layer l alone (mi355_dense_set_layer_window) from the ORACLE's stream and is
    compared with the oracle's stream after the layer -- no amplification through the stack, every launch at full size;
  * the whole step end to end: logits and greedy tokens.
PARITY UNPINNED (no reference fixture exists for these float kernels)."""
import ctypes
import time

import numpy as np

from oracle import dense_llama as DL
from oracle import gptq as G
from oracle import ops as O


def _bf16_bits(a):
    return O.f32_to_bf16_bits(np.ascontiguousarray(a, np.float32))


def ragged_batch32(rng):
    return [4097, 3001, 2049, 1025] + [int(x) for x in rng.integers(64, 257, 28)]


class DensePair:
    def __init__(self, kind, log=None, std=None, seed=77, num_blocks=448, max_batch=None):
        import torch
        from candle_vllm_amd import dense_model as DM
        self.torch, self.DM, self.kind = torch, DM, kind
        self.log = log or (lambda *a: None)
        if kind == "bf16_b32":
            cfg = DL.DenseConfig()                                        # Llama-3-8B (public model card), bf16
            self.max_batch, gptq = 32, False
        elif kind == "gptq_qwen2":
            cfg = DL.DenseConfig(hidden=3584, n_layers=28, n_heads=28, n_kv_heads=4, head_dim=128, intermediate=18944, vocab=152064,
                                 rms_eps=1e-6, rope_theta=1000000.0, qkv_bias=True)
            self.max_batch, gptq = 1, True
        else:
            raise ValueError(kind)
        if max_batch:
            self.max_batch = int(max_batch)
        self.cfg = cfg
        # std: the bench's synthetic weights (0.02 dense; scales 0.002..0.01 GPTQ) make every branch's gain > 1 -- fine for the
        # per-layer checks, chaotic end to end; `std` scales them down to trained-checkpoint-like gains for the end-to-end checks
        rng = np.random.default_rng(seed)
        t0 = time.time()
        H, Hkv, D, hid, I = cfg.n_heads, cfg.n_kv_heads, cfg.head_dim, cfg.hidden, cfg.intermediate
        shapes = {"wq": (H * D, hid), "wk": (Hkv * D, hid), "wv": (Hkv * D, hid), "wo": (hid, H * D), "w1": (I, hid), "w3": (I, hid), "w2": (hid, I)}
        self.max_blocks = -(-(4096 + 64) // cfg.block_size) + 1
        gm = DM.DenseLlama(cfg, max_batch=self.max_batch, max_blocks_per_seq=self.max_blocks, kv_layout=DM.KV_PAGED)
        self.gm = gm
        lib = DM.lib

        def put_dev(layer, name, bits):                                  # bf16 bits (numpy uint16) -> device -> model slot
            t = torch.from_numpy(np.ascontiguousarray(bits).view(np.int16).reshape(-1)).cuda()
            DM._check(lib.mi355_dense_set_weight_dev(gm.h, layer, DM.W_SLOTS[name], t.data_ptr(), t.numel()), name)
            torch.cuda.synchronize()

        def rnd16(shape, s, mean=0.0):
            return DL.R(rng.normal(mean, s, size=shape).astype(np.float32))
        base = rnd16((1002, hid), 0.5)                                    # vocabulary rows repeat a 1002-row base (as the GGUF leg)
        emb = np.ascontiguousarray(np.tile(base, (-(-cfg.vocab // 1002), 1))[: cfg.vocab])
        wstd = std if std is not None else 0.02
        obase = rnd16((1002, hid), wstd)
        out_w = np.ascontiguousarray(np.tile(obase, (-(-cfg.vocab // 1002), 1))[: cfg.vocab])
        W = {"tok_embd": emb, "output": out_w, "output_norm": rnd16(hid, 0.02, 1.0), "layers": []}
        put_dev(-1, "tok_embd", _bf16_bits(emb)); put_dev(-1, "output", _bf16_bits(out_w)); put_dev(-1, "output_norm", _bf16_bits(W["output_norm"]))
        shared = {}
        for name, (n, k) in shapes.items():
            if gptq:
                qw = rng.integers(0, 2 ** 32, (k // 8, n), dtype=np.uint32)
                lo, hi = (0.002, 0.01) if std is None else (0.2 * std, std)
                sc = DL.R(rng.uniform(lo, hi, (k // 128, n)).astype(np.float32))
                shared[name] = {"qweight": qw, "scales": sc, "group": 128}
            else:
                w = rnd16((n, k), wstd)
                shared[name] = {"f64": w.astype(np.float64), "bits": torch.from_numpy(_bf16_bits(w).view(np.int16).reshape(-1)).cuda()}
        for l in range(cfg.n_layers):
            lw = {"attn_norm": rnd16(hid, 0.02, 1.0), "ffn_norm": rnd16(hid, 0.02, 1.0)}
            put_dev(l, "attn_norm", _bf16_bits(lw["attn_norm"])); put_dev(l, "ffn_norm", _bf16_bits(lw["ffn_norm"]))
            if cfg.qkv_bias:
                for bname, n in (("bq", H * D), ("bk", Hkv * D), ("bv", Hkv * D)):
                    lw[bname] = rnd16(n, 0.02)
                    put_dev(l, bname, _bf16_bits(lw[bname]))
            for name in shapes:
                sh = shared[name]
                if gptq:
                    lw[name] = sh                                         # the dict caches its dequantised matrix on first use
                    gm.set_gptq(l, name, sh["qweight"], sh["scales"], 128)
                else:
                    lw[name] = sh["f64"]
                    DM._check(lib.mi355_dense_set_weight_dev(gm.h, l, DM.W_SLOTS[name], sh["bits"].data_ptr(), sh["bits"].numel()), name)
            W["layers"].append(lw)
        torch.cuda.synchronize()
        self.W = W
        self.orc = DL.OracleDenseLlama(cfg, W, flash_layout=False)
        self.log(f"{kind}: weights in both models: {time.time() - t0:.1f}s")
        # ---- KV block pool: random bf16 K/V in every block, one base pool rolled per layer, paged layout on both sides
        t0 = time.time()
        self.num_blocks = num_blocks
        gm.alloc_kv_cache(num_blocks)
        ks, vs = O.kv_cache_shapes(num_blocks, cfg.block_size, Hkv, D, 2, False)
        kb = (rng.standard_normal(ks, dtype=np.float32).view(np.uint32) >> 16).astype(np.uint16)
        vb = (rng.standard_normal(vs, dtype=np.float32).view(np.uint32) >> 16).astype(np.uint16)
        self.cache = []
        for l in range(cfg.n_layers):
            k, v = np.roll(kb, l, axis=0), np.roll(vb, 3 * l + 1, axis=0)
            self.cache.append((k, v))
            gm.kv_upload(l, k, v)
        self.perm = rng.permutation(num_blocks - 1) + 1
        self.rng = rng
        self.log(f"{kind}: KV pool ({num_blocks} blocks) in both models: {time.time() - t0:.1f}s")

    def run(self, seq_lens, layers=None):
        """one decode step of len(seq_lens) sequences whose first seq_len - 1 tokens sit in the cache"""
        cfg, gm, rng, torch, DM = self.cfg, self.gm, self.rng, self.torch, self.DM
        lib = DM.lib
        B, bs, hid = len(seq_lens), cfg.block_size, cfg.hidden
        nxt, seqs = 0, []
        for L in seq_lens:
            n = -(-(int(L) + 1) // bs)
            seqs.append({"tokens": [0] * (int(L) - 1) + [int(rng.integers(0, cfg.vocab))], "block_table": [int(x) for x in self.perm[nxt: nxt + n]]})
            nxt += n
        meta = O.prepare_decode(seqs, bs)
        bt = np.zeros((B, self.max_blocks), np.uint32)
        for i, s in enumerate(seqs):
            bt[i, : len(s["block_table"])] = s["block_table"]
        meta["block_tables"] = bt
        # the GPU step first (the oracle's forward writes the new token's K/V into the shared host pool afterwards; the GPU
        # writes its own into the device pool)
        got = gm.forward(meta).cpu().numpy()
        t0 = time.time()
        trace, mid = [], []
        ref = self.orc.forward(meta, self.cache, trace=trace, trace_mid=mid)
        t_orc = time.time() - t0
        scale = np.abs(ref).max(axis=1)
        err = np.abs(got - ref).max(axis=1)
        tok_g, tok_r = got.argmax(axis=1), ref.argmax(axis=1)
        equal, tie = True, False
        for b in range(B):
            if int(tok_g[b]) != int(tok_r[b]):
                top2 = np.partition(ref[b], -2)[-2:]
                if float(top2[1] - top2[0]) <= 2.0 * float(err[b]):
                    tie = True
                else:
                    equal = False
        res = {"leg": self.kind, "batch": B, "ctx_max": int(max(seq_lens)), "layers": cfg.n_layers, "logits_max_rel_err": float((err / scale).max()),
               "tokens_equal": bool(equal), "near_tie": bool(tie), "oracle": "bf16 rounding chain (unpinned)", "oracle_s": round(t_orc, 1)}
        # ---- every layer alone, from the oracle's stream
        xin = torch.empty((B, hid), dtype=torch.int16, device="cuda")
        xout = torch.empty((B, hid), dtype=torch.int16, device="cuda")
        worst_excess, worst_layer, flips = 0.0, -1, 0.0
        per_layer, worst_ratio = [], 0.0
        try:
            for l in (range(cfg.n_layers) if layers is None else layers):
                xin.copy_(torch.from_numpy(_bf16_bits(trace[l]).view(np.int16)).cuda())
                DM._check(lib.mi355_dense_set_layer_window(gm.h, l, l, xin.data_ptr(), xout.data_ptr()), "layer_window")
                gm.forward(meta)
                g = O.bf16_bits_to_f32(xout.cpu().numpy().view(np.uint16))
                r = trace[l + 1]
                # the stream is a bf16 tensor: an element may sit one ulp (2^-8 of its value) off after any of the layer's rounding
                # points; what counts is the error BEYOND one ulp of the element, relative to the row's largest value
                excess = np.maximum(np.abs(g - r) - 2.0 ** -7 * np.abs(r), 0.0)
                e = float((excess.max(axis=1) / np.abs(r).max(axis=1)).max())
                per_layer.append(round(e, 6))
                flips = max(flips, float((g != r).mean()))
                # the bound of that excess, DERIVED from the oracle's own magnitudes in this layer: the stream is rounded to bf16 where
                # o_proj returns (y1), at the first residual sum (m = x + y1), where down_proj returns (y2) and at the second sum (the
                # element's own ulp, already subtracted above).  Two implementations that differ by f32 summation order alone can land
                # on different sides of a tie at each of those sites, i.e. differ by at most one ulp = 2^-7 of THAT value; the three
                # kicks add up in the worst case: excess_i <= 2^-7 (|y1_i| + |m_i| + |y2_i|), relative to the row's largest |r|.
                y1, m_, y2 = mid[l] - trace[l], mid[l], r - mid[l]
                bound = float(((2.0 ** -7 * (np.abs(y1) + np.abs(m_) + np.abs(y2))).max(axis=1) / np.abs(r).max(axis=1)).max())
                worst_ratio = max(worst_ratio, e / bound)
                if e > worst_excess:
                    worst_excess, worst_layer = e, l
        finally:
            DM._check(lib.mi355_dense_set_layer_window(gm.h, -1, -1, None, None), "layer_window off")
        res.update({"worst_layer_excess": worst_excess, "worst_layer": worst_layer, "max_flip_frac": flips, "per_layer": per_layer,
                    "worst_excess_over_derived_bound": worst_ratio,
                    "units": "worst_layer_excess: error of the stream after a layer beyond one bf16 ulp of the element, relative to the row's largest "
                             "value; max_flip_frac: fraction of stream elements that differ at all (one-ulp flips)"})
        return res
