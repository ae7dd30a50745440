"""Micro-benchmarks of the safetensors-path linears at Llama-3-8B / Qwen2-7B decode shapes (run on the MI355X):
python tests/bench_linear.py [--batch 1] -> us/launch and achieved GB/s of ALGORITHMIC weight bytes."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import candle_vllm_amd.ops as cv  # noqa: E402
from tests.bench_kernels import timeit  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    T = args.batch
    flush = torch.zeros(128 << 20, dtype=torch.float32, device=dev)
    shapes = [("qkv", 6144, 4096, cv.EPI_STORE), ("wo", 4096, 4096, cv.EPI_RESID),
              ("gate_up", 28672, 4096, cv.EPI_SILU_MUL), ("down", 4096, 14336, cv.EPI_RESID),
              ("lm_head", 128256, 4096, cv.EPI_STORE)]
    for dt in (torch.bfloat16,):
        for name, n, k, epi in shapes:
            w = (torch.randn(n, k, device=dev) * 0.02).to(dt)
            x = torch.randn(T, k, device=dev).to(dt)
            n_out = n // 2 if epi == cv.EPI_SILU_MUL else n
            res = torch.randn(T, n_out, device=dev).to(dt) if epi == cv.EPI_RESID else None
            out = torch.empty(T, n_out, device=dev, dtype=dt)
            lin = cv.Linear(w)
            med, best = timeit(lambda: lin.forward(x, epilogue=epi, residual=res, out=out), flush=flush)
            print(f"dense16 {name:8s} T={T:3d} n={n:6d} k={k:6d}  {med:8.1f} us (best {best:7.1f})  "
                  f"{n * k * 2 / med / 1e3:7.1f} GB/s", flush=True)
            del w, lin
        for name, n, k, epi in shapes:
            gs = 128
            qw = torch.randint(-2 ** 31, 2 ** 31 - 1, (k // 8, n), dtype=torch.int32, device=dev)
            sc = (torch.rand(k // gs, n, device=dev) * 0.01 + 0.005).to(dt)
            x = torch.randn(T, k, device=dev).to(dt)
            n_out = n // 2 if epi == cv.EPI_SILU_MUL else n
            res = torch.randn(T, n_out, device=dev).to(dt) if epi == cv.EPI_RESID else None
            out = torch.empty(T, n_out, device=dev, dtype=dt)
            lin = cv.GPTQLinear(qw, sc, gs, scales_permuted=True)
            med, best = timeit(lambda: lin.forward(x, epilogue=epi, residual=res, out=out), flush=flush)
            nbytes = n * k / 2 + (k // gs) * n * 2
            print(f"gptq4   {name:8s} T={T:3d} n={n:6d} k={k:6d}  {med:8.1f} us (best {best:7.1f})  "
                  f"{nbytes / med / 1e3:7.1f} GB/s", flush=True)


if __name__ == "__main__":
    main()
