"""Writes tests/golden/ref_inputs.json: the explicit inputs that oracle/ref_vectors/dump_vectors.rs (a candle-vllm example, run on a box with
Rust) pushes through the reference's own CPU arithmetic.  Everything is spelled out -- quantised blocks as hex bytes, f32 values as decimal
numbers with enough digits to round-trip -- so that no random generator has to agree across languages.  Run from the repository root:
    python oracle/ref_vectors/make_inputs.py
Test infrastructure; nothing of the product imports it."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import kquants as kq          # noqa: E402
from oracle import ops as O               # noqa: E402


def T(a):
    a = np.ascontiguousarray(a, np.float32)
    return {"shape": list(a.shape), "data": [float(np.float32(v)) for v in a.reshape(-1)]}


def main():
    rng = np.random.default_rng(20260925)
    doc = {"qmatmul": [], "rms_norm": [], "silu_mul": [], "rope": [], "attention_bf16": [], "argmax": []}
    for name, t, n, k, tok in (("q4k_8x512", kq.GGML_Q4_K, 8, 512, 3), ("q6k_8x512", kq.GGML_Q6_K, 8, 512, 3),
                               ("q4k_outliers", kq.GGML_Q4_K, 4, 256, 2)):
        w = rng.normal(0, 0.05, (n, k)).astype(np.float32)
        x = rng.normal(0, 1, (tok, k)).astype(np.float32)
        if "outliers" in name:
            x[0, 7] = 300.0
            x[1, :] *= 1e-3
        blocks = kq.quantize(w, t)
        doc["qmatmul"].append({"name": name, "ggml_type": {kq.GGML_Q4_K: "q4_k", kq.GGML_Q6_K: "q6_k"}[t], "n": n, "k": k,
                               "blocks_hex": np.ascontiguousarray(blocks).tobytes().hex(), "x": T(x)})
    x = rng.normal(0, 2, (3, 64)).astype(np.float32)
    doc["rms_norm"].append({"name": "rms_norm_3x64", "x": T(x), "w": T(1 + 0.1 * rng.normal(size=64)), "eps": 1e-5})
    doc["silu_mul"].append({"name": "silu_mul_2x48", "gate": T(rng.normal(0, 3, (2, 48))), "up": T(rng.normal(0, 1, (2, 48)))})
    cos, sin = O.rope_tables(500000.0, 16, 64)
    pos = [5, 63]
    xr = rng.normal(0, 1, (1, 2, len(pos), 16)).astype(np.float32)                 # [b, h, t, d]
    for inter in (True, False):
        doc["rope"].append({"name": "rope_i" if inter else "rope", "interleaved": inter, "x": T(xr),
                            "cos": T(cos[pos]), "sin": T(sin[pos]), "positions": pos})
    H, Hkv, D, Tk = 4, 2, 32, 50
    q = O.round_bf16(rng.normal(0, 1, (1, H, 1, D)).astype(np.float32))
    k = O.round_bf16(rng.normal(0, 1, (1, Hkv, Tk, D)).astype(np.float32))
    v = O.round_bf16(rng.normal(0, 1, (1, Hkv, Tk, D)).astype(np.float32))
    doc["attention_bf16"].append({"name": "naive_attention_bf16", "q": T(q), "k": T(k), "v": T(v), "n_rep": H // Hkv, "scale": 1.0 / np.sqrt(D)})
    a = rng.normal(0, 1, (2, 40)).astype(np.float32)
    a[0, 7] = a[0, 31] = 9.0                                                       # a tie: which index does `argmax` return?
    doc["argmax"].append({"name": "argmax_ties", "x": T(a)})
    out = os.path.join(ROOT, "tests", "golden", "ref_inputs.json")
    with open(out, "w") as f:
        json.dump(doc, f)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
