"""Scratch benchmark (run on the MI355X): BASELINE config 3 shape -- Llama-3-8B bf16 safetensors host path, greedy
decode at batch 1 and 32 (ragged contexts), synthetic weights.  Prints tokens/s and achieved algorithmic GB/s
(weights once per step + live KV)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import dense_llama as DL  # noqa: E402
from candle_vllm_amd import dense_model as M  # noqa: E402


def run(gm, cfg, B, ctxs, steps, warmup):
    bs = cfg.block_size
    nblk = [-(-(c + steps + warmup + 2) // bs) for c in ctxs]
    maxb = max(nblk)
    rng = np.random.default_rng(0)
    ids = rng.permutation(sum(nblk))
    bt = np.zeros((B, maxb), np.int32)
    o = 0
    for i, n in enumerate(nblk):
        bt[i, :n] = ids[o:o + n]
        o += n
    dev = "cuda"
    bt_d = torch.from_numpy(bt).to(dev)
    tok = torch.randint(0, cfg.vocab, (B,), dtype=torch.int32, device=dev)
    logits = torch.empty((B, cfg.vocab), dtype=torch.float32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    lens = np.array(ctxs, np.int64)

    def step():
        pos = torch.from_numpy(lens - 1).to(dev)
        slots = torch.from_numpy(np.array([bt[i, (lens[i] - 1) // bs] * bs + (lens[i] - 1) % bs for i in range(B)], np.int64)).to(dev)
        ctx = torch.from_numpy(lens.astype(np.int32)).to(dev)
        gm.forward_device(tok, pos, slots, bt_d, ctx, int(lens.max()), logits, st)
    for _ in range(warmup):
        step(); lens += 1
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step(); lens += 1
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    kv = float(np.mean(lens)) * B * 2 * cfg.n_layers * cfg.n_kv_heads * cfg.head_dim * 2
    byts = gm.weight_bytes() + kv
    return B / dt, dt * 1e3, byts / dt / 1e9


def main():
    cfg = DL.DenseConfig()
    gm = M.DenseLlama(cfg, max_batch=32, max_blocks_per_seq=80, kv_layout=M.KV_PAGED)
    gm.load_synthetic()
    gm.alloc_kv_cache(32 * 70 + 8)
    rng = np.random.default_rng(1)
    for B, ctxs in ((1, [4096]), (32, rng.integers(256, 4097, 32).tolist())):
        tps, ms, gbs = run(gm, cfg, B, ctxs, steps=16, warmup=3)
        print(f"dense bf16 Llama-3-8B  B={B:2d}  {tps:8.1f} tok/s  {ms:7.3f} ms/step  {gbs:7.1f} GB/s algorithmic", flush=True)


if __name__ == "__main__":
    main()
