"""__graft_entry__.smoke(): one tiny decode step of the hot path on cuda:0 (tiny GGUF-llama shapes, Q4_K/Q6_K
weights, paged KV, hipGraph-less eager step through the C ABI), checked against the CPU oracle."""
import numpy as np


def run_smoke():
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("smoke() needs cuda:0 (MI355X)")
    torch.cuda.set_device(0)
    from oracle import llama
    from oracle import ops as O
    from candle_vllm_amd import model as M
    cfg = llama.LlamaConfig.tiny()
    W = llama.make_weights(cfg, seed=1234)
    orc = llama.OracleLlama(cfg, W)
    rng = np.random.default_rng(7)
    seqs = [{"tokens": [int(t) for t in rng.integers(0, cfg.vocab, 19)], "block_table": [3, 7]},
            {"tokens": [int(t) for t in rng.integers(0, cfg.vocab, 5)], "block_table": [1]}]
    cache = orc.new_cache(16)
    lg = orc.forward(O.prepare_prompt(seqs, cfg.block_size), cache, is_prefill=True)
    for s, row in zip(seqs, lg):
        s["tokens"].append(int(row.argmax()))
    gm = M.GGUFLLaMa(cfg, max_batch=2)
    gm.load_oracle_weights(W)
    gm.alloc_kv_cache(16)
    for l, (kc, vc) in enumerate(cache):
        gm.kv_upload(l, kc, vc)
    meta = O.prepare_decode(seqs, cfg.block_size)
    got = gm.forward_decode(meta).cpu().numpy()
    ref = orc.forward(meta, cache)
    err = float(np.abs(got - ref).max() / np.abs(ref).max())
    if not err < 1e-3:
        raise AssertionError(f"smoke: logits rel err {err} >= 1e-3")
    if [int(r.argmax()) for r in got] != [int(r.argmax()) for r in ref]:
        raise AssertionError("smoke: greedy tokens differ from the oracle")
    print(f"smoke ok: decode step logits rel err {err:.2e}, tokens {[int(r.argmax()) for r in got]}")
