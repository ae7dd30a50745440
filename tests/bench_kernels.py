"""Kernel micro-benchmarks at Llama-3-8B decode shapes (run on the MI355X):
python tests/bench_kernels.py [--batch 1] -> one line per kernel with us/launch and achieved GB/s of
ALGORITHMIC bytes.  Scratch tool for tuning; bench.py is the judged benchmark."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import candle_vllm_amd.ops as cv  # noqa: E402


def rand_tiles(ggml_type, n, k, device):
    """random VALID repacked weights directly on the GPU (random codes/scales, sane f16 d/dmin)"""
    nb = cv.lib.mi355_qweight_repacked_size(ggml_type, n, k)
    t = torch.randint(0, 256, (nb,), dtype=torch.uint8, device=device)
    ntile = (n + 15) // 16 * (k // 256)
    lo, hi = (int(b) for b in np.array([0.001], np.float16).view(np.uint8))
    if ggml_type == cv.GGML_Q4_K:
        tv = t.view(ntile, 2304)
        tv[:, 0:256:16] = lo
        tv[:, 1:256:16] = hi
        tv[:, 2:256:16] = lo
        tv[:, 3:256:16] = hi
    else:
        tv = t.view(ntile, 3360)
        tv[:, 3328:3360:2] = lo
        tv[:, 3329:3360:2] = hi
    return t


class FakeMat:
    def __init__(self, t, n, k, device):
        self.ggml_type, self.n, self.k = t, n, k
        self.tiles = rand_tiles(t, n, k, device)
        self.bytes = self.tiles.numel()


def timeit(fn, iters=30, warmup=3, flush=None):
    """Enqueue everything first (host runs ahead of the GPU), then read the event pairs: the measured
    span is kernel time + dispatch gap, not Python launch latency."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    evs = []
    for _ in range(iters):
        if flush is not None:
            flush.add_(1.0)                   # 512 MB sweep: evicts L2 + Infinity Cache between launches
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        evs.append((s, e))
    torch.cuda.synchronize()
    times = sorted(s.elapsed_time(e) * 1e3 for s, e in evs)
    return times[len(times) // 2], times[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--ctx", type=int, default=4608)
    ap.add_argument("--nw", type=int, default=0)
    ap.add_argument("--r", type=int, default=0)
    ap.add_argument("--dbg", type=int, default=0)
    ap.add_argument("--only-qmm", action="store_true")
    args = ap.parse_args()
    dev = "cuda"
    cv.lib.mi355_set_tuning(0, args.nw)
    cv.lib.mi355_set_tuning(1, args.r)
    cv.lib.mi355_set_tuning(2, args.dbg)
    print(f"tuning nw={args.nw} r={args.r} dbg={args.dbg}", flush=True)
    B = args.batch
    hid, I, H, Hkv, D, V = 4096, 14336, 32, 8, 128, 128256
    flush = torch.zeros(128 * 1024 * 1024, dtype=torch.float32, device=dev)
    x = torch.randn(B, hid, device=dev)
    nw = torch.ones(hid, device=dev)
    res = []

    def report(name, med, mn, nbytes):
        print(f"{name:28s} B={B:2d}  median {med:8.1f} us  min {mn:8.1f} us  {nbytes / 1e6:8.1f} MB  "
              f"{nbytes / med / 1e3:7.1f} GB/s (median)  {nbytes / mn / 1e3:7.1f} GB/s (best)", flush=True)
        res.append((name, med, mn, nbytes))

    # --- plain matmuls
    for name, t, n, k in (("wq  q4k 4096x4096", 12, hid, hid), ("wo  q4k 4096x4096", 12, hid, hid),
                          ("down q4k 4096x14336", 12, hid, I), ("down q6k 4096x14336", 14, hid, I),
                          ("lm_head q6k 128256x4096", 14, V, hid)):
        m = FakeMat(t, n, k, dev)
        xin = torch.randn(B, k, device=dev)
        out = torch.empty(B, n, device=dev)
        f = lambda: cv._check(cv.lib.mi355_qmatmul(out.data_ptr(), xin.data_ptr(), m.tiles.data_ptr(), t, B, n, k,
                                                   None, cv._stream()), "qmm")
        med, mn = timeit(f, flush=flush)
        report(name, med, mn, m.bytes)
        del m
    # --- fused norm + qkv + rope + cache
    mq, mk, mv = FakeMat(12, H * D, hid, dev), FakeMat(12, Hkv * D, hid, dev), FakeMat(14, Hkv * D, hid, dev)
    NB = B * (-(-(args.ctx + 1) // 64)) + 1
    kc = torch.randn(NB, 64, Hkv, D, device=dev).to(torch.bfloat16)
    vc = torch.randn(NB, 64, Hkv, D, device=dev).to(torch.bfloat16)
    cos = torch.randn(8192, D // 2, device=dev)
    sin = torch.randn(8192, D // 2, device=dev)
    pos = torch.full((B,), args.ctx, dtype=torch.int64, device=dev)
    nblk = -(-(args.ctx + 1) // 64)
    bt = (torch.arange(B * nblk, dtype=torch.int32, device=dev).reshape(B, nblk) + 1).contiguous()
    slots = (bt[:, args.ctx // 64].to(torch.int64) * 64 + args.ctx % 64).contiguous()
    q_out = torch.empty(B, H * D, dtype=torch.bfloat16, device=dev)
    rope = dict(cos=cos, sin=sin, positions=pos, slot_mapping=slots, q_out=q_out, key_cache=kc, value_cache=vc,
                num_heads=H, num_kv_heads=Hkv, head_dim=D)
    f = lambda: cv.qmatmul_fused([mq, mk, mv], x, epilogue=cv.EPI_QKV_ROPE_CACHE, norm_weight=nw, norm_eps=1e-5, rope=rope)
    med, mn = timeit(f, flush=flush)
    report("norm+qkv+rope+cache", med, mn, mq.bytes + mk.bytes + mv.bytes)
    # --- fused norm + gate/up + silu
    mg, mu = FakeMat(12, I, hid, dev), FakeMat(12, I, hid, dev)
    h = torch.empty(B, I, device=dev)
    f = lambda: cv.qmatmul_fused([mg, mu], x, epilogue=cv.EPI_SILU_MUL, out=h, norm_weight=nw, norm_eps=1e-5)
    med, mn = timeit(f, flush=flush)
    report("norm+gate/up+silu", med, mn, mg.bytes + mu.bytes)
    if args.only_qmm:
        return
    # --- paged attention
    pa = cv.PagedAttention(H, D, D ** -0.5, Hkv)
    cl = torch.full((B,), args.ctx + 1, dtype=torch.int32, device=dev)
    meta = cv.InputMetadata(False, slots, bt, cl, max_context_len=args.ctx + 1)
    kv_bytes = B * (args.ctx + 1) * 2 * Hkv * D * 2
    for ps in (0, 128, 256, 512, None):
        f = lambda: pa.decode(q_out.view(B, H, D), kc, vc, meta, None, partition_size=ps)
        med, mn = timeit(f, flush=flush)
        auto = cv.choose_partition(B, Hkv, args.ctx + 1)
        report(f"paged_attn ps={'auto(%d)' % auto if ps is None else ps}", med, mn, kv_bytes)
    # vLLM (paged) layout: MFMA kernel
    kcp = torch.randn(NB, Hkv, D // 8, 64, 8, device=dev).to(torch.bfloat16)
    vcp = torch.randn(NB, Hkv, D, 64, device=dev).to(torch.bfloat16)
    for ps in (32, 64, 128):
        f = lambda: pa.decode(q_out.view(B, H, D), kcp, vcp, meta, None, partition_size=ps)
        med, mn = timeit(f, flush=flush)
        report(f"paged_attn MFMA ps={ps}", med, mn, kv_bytes)
    # --- small ops
    f = lambda: cv.rms_norm(x, nw, 1e-5)
    med, mn = timeit(f)
    report("rms_norm", med, mn, B * hid * 8)
    tot = sum(r[3] for r in res)
    print("done", flush=True)


if __name__ == "__main__":
    main()
