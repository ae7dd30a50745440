"""Scratch stress tool (run on the MI355X): repeat the same tiny-model decode step many times and check the logits
are bit-identical -- hunts for races in the kernels at the tiny shapes the unit tests use."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import llama, ops as O  # noqa: E402
from candle_vllm_amd import model as M  # noqa: E402

cfg = llama.LlamaConfig.tiny()
W = llama.make_weights(cfg, seed=1234)
rng = np.random.default_rng(7)
for flash in (True, False):
    for nseq in (1, 2, 3, 4):
        seqs = [{"tokens": [int(t) for t in rng.integers(0, cfg.vocab, 5 + 7 * i)], "block_table": [1 + 2 * i, 2 + 2 * i]}
                for i in range(nseq)]
        gm = M.GGUFLLaMa(cfg, max_batch=4, kv_layout=M.KV_FLASH if flash else M.KV_PAGED)
        gm.load_oracle_weights(W)
        gm.alloc_kv_cache(16)
        ks, vs = O.kv_cache_shapes(16, cfg.block_size, cfg.n_kv_heads, cfg.head_dim, 2, flash)
        kc = O.f32_to_bf16_bits(rng.normal(0, 1, ks).astype(np.float32))
        vc = O.f32_to_bf16_bits(rng.normal(0, 1, vs).astype(np.float32))
        for l in range(cfg.n_layers):
            gm.kv_upload(l, kc, vc)
        meta = O.prepare_decode(seqs, cfg.block_size)
        ref = gm.forward_decode(meta).cpu().numpy()
        bad = 0
        for it in range(300):
            got = gm.forward_decode(meta).cpu().numpy()
            if not np.array_equal(got, ref):
                bad += 1
                if bad <= 3:
                    d = np.abs(got - ref)
                    print("  MISMATCH it", it, "rows", np.unique(np.nonzero(d)[0]).tolist(), "max", d.max(), flush=True)
        print(f"flash={flash} nseq={nseq}: {bad} / 300 differ", flush=True)

# ---- wide (9..32 token) chained path + workgroup-merged attention partitions: batch 12, contexts up to 600 tokens
B = 12
ctxs = [int(c) for c in rng.integers(100, 600, B)]
nblk = [-(-(c + 1) // cfg.block_size) for c in ctxs]
seqs, nxt = [], 1
for c, n in zip(ctxs, nblk):
    seqs.append({"tokens": [int(t) for t in rng.integers(0, cfg.vocab, c)], "block_table": list(range(nxt, nxt + n))})
    nxt += n
big = llama.LlamaConfig.tiny(max_seq=1024)
gm = M.GGUFLLaMa(big, max_batch=B, max_blocks_per_seq=max(nblk) + 1, kv_layout=M.KV_PAGED)
gm.load_oracle_weights(W)
gm.alloc_kv_cache(nxt + 1)
ks, vs = O.kv_cache_shapes(nxt + 1, cfg.block_size, cfg.n_kv_heads, cfg.head_dim, 2, False)
kc = O.f32_to_bf16_bits(rng.normal(0, 1, ks).astype(np.float32))
vc = O.f32_to_bf16_bits(rng.normal(0, 1, vs).astype(np.float32))
for l in range(cfg.n_layers):
    gm.kv_upload(l, kc, vc)
meta = O.prepare_decode(seqs, cfg.block_size)
ref = gm.forward_decode(meta).cpu().numpy()
bad = 0
for it in range(300):
    got = gm.forward_decode(meta).cpu().numpy()
    if not np.array_equal(got, ref):
        bad += 1
print(f"batch {B} wide+chained+merged-attention: {bad} / 300 runs differ (max ctx {max(ctxs)})", flush=True)
assert bad == 0
