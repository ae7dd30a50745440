"""Generates tests/golden/tiny_llama_logits.json from the numpy oracle (there are no reference fixtures
for this path -- SURVEY.md section 0.5 -- so the golden vector pins OUR oracle against regressions and lets
the GPU box check it without /root/reference).  Run: python -m tests.golden.make_golden"""
import json
import os

import numpy as np

from oracle import llama
from oracle import ops as O


def main():
    cfg = llama.LlamaConfig.tiny()
    W = llama.make_weights(cfg, seed=1234)
    M = llama.OracleLlama(cfg, W)
    rng = np.random.default_rng(7)
    seqs = [{"tokens": [int(t) for t in rng.integers(0, cfg.vocab, 19)], "block_table": [3, 7]},
            {"tokens": [int(t) for t in rng.integers(0, cfg.vocab, 5)], "block_table": [1]}]
    cache = M.new_cache(16)
    lg = M.forward(O.prepare_prompt(seqs, cfg.block_size), cache, is_prefill=True)
    for s, row in zip(seqs, lg):
        s["tokens"].append(int(row.argmax()))
    dec = M.forward(O.prepare_decode(seqs, cfg.block_size), cache)
    out = {"next_tokens": [int(r.argmax()) for r in dec],
           "logits_head": [[float(v) for v in r[:16]] for r in dec]}
    path = os.path.join(os.path.dirname(__file__), "tiny_llama_logits.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
