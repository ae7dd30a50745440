"""Prompt-step (T = 2048) GPTQ-128 projections at Qwen2-7B shapes over the tiled 4-bit image: the one-pass MFMA GEMM against the decode
kernel run in 64-token chunks (tuning key 39); useful TFLOP/s = 2 T N K / time."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from candle_vllm_amd import ops as cv, tuning  # noqa: E402

T = int(os.environ.get("T", "2048"))
g = torch.Generator(device="cuda").manual_seed(1)
for name, N, K, epi in (("q|k|v-like", 4608, 3584, cv.EPI_STORE), ("wo", 3584, 3584, cv.EPI_STORE), ("gate/up+silu", 2 * 18944, 3584, cv.EPI_SILU_MUL),
                        ("down", 3584, 18944, cv.EPI_STORE)):
    qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (K // 8, N), dtype=torch.int32, device="cuda", generator=g)
    sc = (torch.rand((K // 128, N), device="cuda", generator=g) * 0.008 + 0.002).to(torch.bfloat16)
    x = torch.randn((T, K), device="cuda", generator=g).to(torch.bfloat16)
    lin = cv.GPTQLinear(qw, sc, 128)
    res = {}
    for off in (0, 1):
        with tuning(30, 16 if off else 0):     # key 30 bit 16: the one-pass GPTQ prompt GEMM off (chunks of the decode kernel)
            y = lin.forward(x, epilogue=epi)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5):
                y = lin.forward(x, epilogue=epi)
            torch.cuda.synchronize()
            res[off] = ((time.perf_counter() - t0) / 5, y.float())
    d = float((res[0][1] - res[1][1]).abs().max() / res[1][1].abs().max())
    print(f"{name:14s} T={T} N={N} K={K}: one-pass GEMM {res[0][0] * 1e6:9.1f} us ({2.0 * T * N * K / res[0][0] / 1e12:7.1f} TFLOP/s)   "
          f"decode kernel x {T // 64} chunks {res[1][0] * 1e6:9.1f} us   max rel diff {d:.2e}", flush=True)
