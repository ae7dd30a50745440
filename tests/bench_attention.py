"""Scratch micro-benchmark (run on the MI355X): MFMA paged-layout decode attention at batch 32 -- uniform vs ragged
context lengths, fused merge on/off.  us/launch and GB/s of the live KV bytes."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import candle_vllm_amd.ops as cv  # noqa: E402
from tests.bench_kernels import timeit  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    B, H, Hkv, D, bs = 32, 32, 8, 128, 64
    flush = torch.zeros(128 << 20, dtype=torch.float32, device=dev)
    rng = np.random.default_rng(4321)
    cases = {"uniform 4096": np.full(B, 4097), "uniform 2100": np.full(B, 2100),
             "ragged U[256,4096]": rng.integers(256, 4097, B), }
    cases["ragged sorted desc"] = np.sort(cases["ragged U[256,4096]"])[::-1].copy()
    maxb = 4097 // bs + 2
    NB = B * maxb + 1
    kcp = torch.randn(NB, Hkv, D // 8, bs, 8, device=dev).to(torch.bfloat16)
    vcp = torch.randn(NB, Hkv, D, bs, device=dev).to(torch.bfloat16)
    q = torch.randn(B, H, D, device=dev).to(torch.bfloat16)
    perm = rng.permutation(NB - 1) + 1
    bt = torch.from_numpy(perm[: B * maxb].reshape(B, maxb).astype(np.int32)).to(dev)
    pa = cv.PagedAttention(H, D, D ** -0.5, Hkv)
    for name, ctx in cases.items():
        cl = torch.from_numpy(ctx.astype(np.int32)).to(dev)
        meta = cv.InputMetadata(False, None, bt, cl, max_context_len=int(ctx.max()))
        kv_bytes = float(ctx.sum()) * 2 * Hkv * D * 2
        for fused in (1, 0):
            cv.lib.mi355_set_tuning(3, fused)
            for ps in (128, 64):
                f = lambda: pa.decode(q, kcp, vcp, meta, None, partition_size=ps)
                med, mn = timeit(f, flush=flush)
                print(f"{name:22s} fused={fused} ps={ps:3d}  {med:7.1f} us  {kv_bytes / med / 1e3:7.1f} GB/s", flush=True)
    cv.lib.mi355_set_tuning(3, 1)


if __name__ == "__main__":
    main()
