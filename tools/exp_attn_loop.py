"""scratch: decode attention alone on the PAGED layout, batch 32 (uniform 4 k and ragged contexts): one-partition waves (32 / 64 tokens)
against the looped chunks (256 / 512, pa_mfma_chunk), the LDS-DMA experiment (1024 / 2048, tuning key 44 = 2) and the generic kernel
(tuning key 44 = 0)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from candle_vllm_amd import ops as cv                 # noqa: E402
from candle_vllm_amd import tuning                      # noqa: E402

B, H, Hkv, D, bs = 32, 32, 8, 128, 64
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(3)
for name, ctx in (("uniform 4096", [4096] * B), ("ragged U[256,4096]", np.random.default_rng(4321).integers(256, 4097, B).tolist())):
    nblk = [-(-c // bs) for c in ctx]
    NB = sum(nblk) + 1
    kc = torch.randn((NB, Hkv, D // 8, bs, 8), device=dev, generator=g).to(torch.bfloat16)
    vc = torch.randn((NB, Hkv, D, bs), device=dev, generator=g).to(torch.bfloat16)
    q = torch.randn((B, H, D), device=dev, generator=g).to(torch.bfloat16)
    perm = np.random.default_rng(5).permutation(NB - 1) + 1
    bt = np.zeros((B, max(nblk)), np.int32)
    o = 0
    for b, n in enumerate(nblk):
        bt[b, :n] = perm[o:o + n]
        o += n
    meta = cv.InputMetadata(False, torch.zeros(B, dtype=torch.int64, device=dev), torch.from_numpy(bt).to(dev),
                            torch.tensor(ctx, dtype=torch.int32, device=dev), max_context_len=max(ctx))
    pa = cv.PagedAttention(H, D, 1 / np.sqrt(D), Hkv)
    kv_bytes = sum(ctx) * Hkv * D * 2 * 2
    ref = None
    for label, ps, loop in (("one-partition waves, 32", 32, 1), ("one-partition waves, 64", 64, 1), ("looped chunks, 256", 256, 1),
                            ("looped chunks, 512", 512, 1), ("LDS-DMA stages, 1024", 1024, 2), ("LDS-DMA stages, 2048", 2048, 2), ("LDS-DMA stages, 4096", 4096, 2), ("LDS-DMA balanced stream", 64, 3), ("LDS-DMA balanced stream, fused merge", 64, 4),
                            ("generic kernel, 256", 256, 0)):
        with tuning(44, loop):
            out = pa.decode(q, kc, vc, meta, None, partition_size=ps)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                pa.decode(q, kc, vc, meta, None, partition_size=ps)
            e1.record()
            torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        if ref is None:
            ref = out.float()
        err = (out.float() - ref).abs().max().item()
        print(f"{name}: {label}: {us:.1f} us per launch, {kv_bytes / us / 1e6:.2f} TB/s of K+V, max |diff| to the first {err:.2e}", flush=True)
