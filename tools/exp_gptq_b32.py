"""Per-launch time of the 4-bit (GPTQ, tiled image) projections of a batch-32 decode step, Qwen2-7B shapes, and of the same launch
with 4 x the rows (how much of a launch is its ramp and tail): python tools/exp_gptq_b32.py  (on the GPU box)."""
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import candle_vllm_amd.ops as cv  # noqa: E402


def bench(name, T, N, K, epi, resid=False, bias=False, iters=40, group=128):
    g = torch.Generator(device="cuda").manual_seed(1)
    qw = torch.randint(0, 2 ** 31 - 1, (K // 8, N), dtype=torch.int32, device="cuda", generator=g)
    sc = (torch.rand((K // group, N), device="cuda", generator=g) * 0.01 + 0.005).to(torch.bfloat16)
    b = torch.zeros(N, dtype=torch.bfloat16, device="cuda") if bias else None
    lin = cv.GPTQLinear(qw, sc, group, bias=b)
    x = torch.randn((T, K), device="cuda", generator=g).to(torch.bfloat16)
    n_out = N // 2 if epi == cv.EPI_SILU_MUL else N
    out = torch.empty((T, n_out), dtype=torch.bfloat16, device="cuda")
    res = torch.zeros((T, n_out), dtype=torch.bfloat16, device="cuda") if resid else None
    flush = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
    kw = dict(epilogue=epi, out=out)
    if resid:
        kw["residual"] = res
    for _ in range(3):
        lin.forward(x, **kw)
    ts = []
    for _ in range(iters):
        flush.zero_()                                           # the weights leave the caches between launches, as inside a step
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        lin.forward(x, **kw)
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    us = ts[len(ts) // 2]
    mb = K * N / 2 / 1e6
    print(f"{name:28s} T={T:3d} N={N:6d} K={K:5d}  {us:8.1f} us  {mb:7.1f} MB  {mb / us:6.2f} TB/s", flush=True)


if __name__ == "__main__":
    T = int(os.environ.get("T", 32))
    if os.environ.get("ONLY") == "ksweep":                      # fixed cost of a launch vs cost per k-block
        for K in (1792, 3584, 7168, 14336):
            bench("gate/up, K sweep", T, 2 * 18944, K, cv.EPI_SILU_MUL)
        for N in (18944 // 4, 18944 // 2, 18944, 2 * 18944):
            bench("gate/up, N sweep", T, 2 * N, 3584, cv.EPI_SILU_MUL)
        sys.exit(0)
    bench("gate/up (SiLU*up)", T, 2 * 18944, 3584, cv.EPI_SILU_MUL)
    bench("gate/up x4 rows", T, 8 * 18944, 3584, cv.EPI_SILU_MUL)
    if os.environ.get("ONLY") == "gateup":
        sys.exit(0)
    bench("down (+resid)", T, 3584, 18944, cv.EPI_RESID, resid=True)
    bench("down x4 rows", T, 4 * 3584, 18944, cv.EPI_RESID, resid=True)
    bench("wo (+resid)", T, 3584, 3584, cv.EPI_RESID, resid=True)
    bench("q|k|v as one matrix (+bias)", T, 4608, 3584, cv.EPI_STORE, bias=True)
