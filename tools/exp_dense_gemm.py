"""hand-written dense 16-bit prompt GEMM (csrc/dense_gemv.hip dense_gemm_kernel): TFLOP/s at Llama-3-8B prompt shapes"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from candle_vllm_amd import ops as cv

dev = "cuda"
T = int(os.environ.get("T", "2048"))
for name, N, K, epi in (("qkv-like", 6144, 4096, cv.EPI_STORE), ("wo+resid", 4096, 4096, cv.EPI_RESID), ("gate_up silu", 28672, 4096, cv.EPI_SILU_MUL),
                        ("down+resid", 4096, 14336, cv.EPI_RESID), ("lm_head", 128256, 4096, cv.EPI_STORE)):
    w = (torch.randn((N, K), device=dev) * 0.02).to(torch.bfloat16)
    x = torch.randn((T, K), device=dev).to(torch.bfloat16)
    lin = cv.Linear(w)
    res = torch.randn((T, N), device=dev).to(torch.bfloat16) if epi == cv.EPI_RESID else None
    kw = {"epilogue": epi}
    if res is not None:
        kw["residual"] = res
    y = lin.forward(x, **kw)
    torch.cuda.synchronize()
    # against torch's own bf16 matmul (f32 accumulate, one rounding) on a sample of rows
    ref = (x[:64].float() @ w.float().T)
    if epi == cv.EPI_SILU_MUL:
        g, u = ref[:, : N // 2].bfloat16().float(), ref[:, N // 2:].bfloat16().float()
        ref = (torch.nn.functional.silu(g).bfloat16().float() * u)
    elif epi == cv.EPI_RESID:
        ref = ref.bfloat16().float() + res[:64].float()
    err = float((y[:64].float() - ref.bfloat16().float()).abs().max() / ref.abs().max())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(5):
        e0.record()
        for _ in range(4):
            lin.forward(x, **kw)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 4)
    print(f"{name:14s} T={T} N={N} K={K}: {best * 1e3:8.1f} us  {2.0 * T * N * K / best / 1e9:7.1f} TFLOP/s  max rel diff vs torch {err:.2e}", flush=True)
