"""Scratch benchmark (run on the MI355X): GGUF Llama-3-8B Q4_K_M prompt step, synthetic weights -- tokens/s of one
`forward_prefill` over a T-token prompt with the GEMM path on / off."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import llama, ops as O  # noqa: E402
from candle_vllm_amd import model as M  # noqa: E402


def main():
    cfg = llama.LlamaConfig.llama3_8b()
    gm = M.GGUFLLaMa(cfg, max_batch=4, max_blocks_per_seq=80, kv_layout=M.KV_PAGED)
    gm.load_synthetic()
    gm.alloc_kv_cache(160)
    for kv in filter(None, os.environ.get("MI355_TUNING", "").split(",")):     # "key:value,..." for the whole run (e.g. 48:1)
        k, v = kv.split(":")
        M.lib.mi355_set_tuning(int(k), int(v))
    rng = np.random.default_rng(0)
    for T in [int(t) for t in os.environ.get("PF_T", "512,2048,4096").split(",")]:
        seqs = [{"tokens": rng.integers(0, cfg.vocab, T).tolist(), "block_table": list(range(1, 1 + -(-T // cfg.block_size)))}]
        meta = O.prepare_prompt(seqs, cfg.block_size)
        for mode in [int(m) for m in os.environ.get("PF_MODES", "1,2,0").split(",")]:   # 1 = hand-written quantised GEMM, 2 = library GEMMs (first generation), 0 = streaming
            if mode == 0 and T > 512:
                continue
            M.lib.mi355_set_tuning(6, mode)
            for qv in ([int(v) for v in os.environ.get("PF_QPG", "2").split(",")] if mode == 1 else [0]):
                M.lib.mi355_set_tuning(11, qv)                      # (probe builds: the product library ignores keys 11 and 2)
                for dbg in ([int(v) for v in os.environ.get("PF_DBG", "0").split(",")] if mode == 1 else [0]):
                    M.lib.mi355_set_tuning(2, dbg)
                    # PF_ATTN="0,1": A/B of the prompt attention on the same model (tuning key 47: 1 = the LDS-ring kernel, the default; 0 = register-fed)
                    for attn in [int(v) for v in os.environ.get("PF_ATTN", "1").split(",")]:
                        M.lib.mi355_set_tuning(47, attn)
                        gm.forward_prefill(meta)
                        t0 = time.perf_counter()
                        gm.forward_prefill(meta)
                        dt = time.perf_counter() - t0
                        flops = 2.0 * (gm.weight_bytes / 0.5625) * T          # ~ 2 * params * tokens (Q4_K: 0.5625 B / weight)
                        print(f"prefill T={T:5d} gemm={mode} variant={qv} dbg={dbg} attn={attn}: {T / dt:9.1f} tok/s  {dt * 1e3:8.1f} ms  ~{flops / dt / 1e12:6.1f} TFLOP/s", flush=True)
                    M.lib.mi355_set_tuning(47, 1)
    M.lib.mi355_set_tuning(11, 2)
    M.lib.mi355_set_tuning(2, 0)
    M.lib.mi355_set_tuning(6, 1)


if __name__ == "__main__":
    main()
