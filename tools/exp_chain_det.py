"""scratch: is the 12-token wide step deterministic run to run, chained and unchained?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import llama
from oracle import ops as O
from candle_vllm_amd import model as M
cfg = llama.LlamaConfig.tiny()
W = llama.make_weights(cfg, seed=77)
orc = llama.OracleLlama(cfg, W, flash_layout=False)
rng = np.random.default_rng(11)
B = 12
seqs = [{"tokens": [int(t) for t in rng.integers(0, cfg.vocab, int(n))], "block_table": [2 * i + 1, 2 * i + 2]}
        for i, n in enumerate(rng.integers(3, 2 * cfg.block_size - 4, B))]
cache = orc.new_cache(2 * B + 2)
lg = orc.forward(O.prepare_prompt(seqs, cfg.block_size), cache, is_prefill=True)
for s, row in zip(seqs, lg):
    s["tokens"].append(int(row.argmax()))
gm = M.GGUFLLaMa(cfg, max_batch=B, kv_layout=M.KV_PAGED)
gm.load_oracle_weights(W)
gm.alloc_kv_cache(2 * B + 2)
meta = O.prepare_decode(seqs, cfg.block_size)
ref = orc.forward(meta, [(k.copy(), v.copy()) for k, v in cache])
outs = []
for chain in (1, 1, 0, 0, 1, 0):
    for l, (kc, vc) in enumerate(cache):
        gm.kv_upload(l, kc, vc)
    M.lib.mi355_set_tuning(9, chain)
    outs.append((chain, gm.forward_decode(meta).cpu().numpy()))
def rel(a, b): return float(np.abs(a - b).max() / np.abs(b).max())
for i, (c, o) in enumerate(outs):
    print(i, "chain", c, "vs ref", rel(o, ref), "vs run0", rel(o, outs[0][1]), "rows differing from run0:", int((o != outs[0][1]).any(axis=1).sum()), flush=True)
