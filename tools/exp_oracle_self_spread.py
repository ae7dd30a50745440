"""CPU-only experiment: how far do two restatements of the reference's CPU decode step drift apart through 32 layers when they differ ONLY in the
summation order / precision of the quantised mat-vecs (O1: f64 dots; O1f: f32 blocked dots; ~1e-6 apart per product), both with the reference's
bf16 attention tensors (models/mod.rs:1288-1306)?  If that alone exceeds 1e-3 of the logit scale, north_star's end-to-end bar is below the
reproducibility of the reference arithmetic itself (two thread counts of candle's CPU backend differ by as much), and no implementation
can be held to it end to end.  Llama-3-8B Q4_K_M shapes, ctx 4096, the full-size parity leg's weights (fill_scale 0.2) and KV pool."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import cref, ops as O
from oracle.llama import LlamaConfig, q4km_type_for

cfg = LlamaConfig.llama3_8b()
names = ["wq", "wk", "wv", "wo", "w1", "w2", "w3"]
cref.build()
types = [q4km_type_for(n, l, cfg.n_layers) for l in range(cfg.n_layers) for n in names] + [q4km_type_for("output", 0, cfg.n_layers)]
t0 = time.time()
orc = cref.CLlama(cfg, W=None, types=types, seed=1235, fill_scale=float(os.environ.get("FILL", "0.2")))
rng = np.random.default_rng(1235)
base = (rng.standard_normal((1002, cfg.hidden)) * 0.02).astype(np.float32)
emb = np.ascontiguousarray(np.tile(base, (-(-cfg.vocab // 1002), 1))[: cfg.vocab])
orc.set_f32(-1, 9, emb)
orc.set_f32(-1, 10, 1.0 + rng.normal(0, 0.02, cfg.hidden))
for l in range(cfg.n_layers):
    orc.set_f32(l, 7, 1.0 + rng.normal(0, 0.02, cfg.hidden))
    orc.set_f32(l, 8, 1.0 + rng.normal(0, 0.02, cfg.hidden))
nb = 80
shape = (nb, cfg.block_size, cfg.n_kv_heads, cfg.head_dim)
kb = (rng.standard_normal(shape, dtype=np.float32).view(np.uint32) >> 16).astype(np.uint16)
vb = (rng.standard_normal(shape, dtype=np.float32).view(np.uint32) >> 16).astype(np.uint16)
print(f"model built in {time.time() - t0:.1f}s", flush=True)
L = 4097
seq = {"tokens": [0] * (L - 1) + [4242], "block_table": list(range(1, 1 + -(-(L + 1) // cfg.block_size)))}
meta = O.prepare_decode([seq], cfg.block_size)
res = {}
for bf in (0, 1):
    cref.lib().orc_llama_set_attn_bf16(bf)
    out = {}
    for o2 in (0, 2):
        cache = [(np.roll(kb, l, axis=0).copy(), np.roll(vb, 3 * l + 1, axis=0).copy()) for l in range(cfg.n_layers)]
        t0 = time.time()
        out[o2] = orc.decode(meta, cache, o2=o2)[0]
        print(f"attn_bf16={bf} o2={o2}: {time.time() - t0:.1f}s", flush=True)
    d = float(np.abs(out[0] - out[2]).max() / np.abs(out[0]).max())
    res[bf] = d
    print(f"attn_bf16={bf}: |O1 - O1f| / max|O1| = {d:.3e}  (same tokens: {int(out[0].argmax()) == int(out[2].argmax())})", flush=True)
cref.lib().orc_llama_set_attn_bf16(0)
