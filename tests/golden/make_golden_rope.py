"""Golden RoPE tables (a few rows) from the numpy restatement of ScalingRotaryEmbedding::new
(src/openai/models/layers/rotary_emb.rs:107-341): python tests/golden/make_golden_rope.py -> tests/golden/rope_tables.json"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import ops as O  # noqa: E402

CASES = {
    "default_theta5e5_d128": dict(theta=5e5, dim=128, max_seq=64, scaling=None, mpe=0),
    "llama3.1": dict(theta=5e5, dim=128, max_seq=64, mpe=131072,
                     scaling={"rope_type": "llama3", "factor": 8.0, "low_freq_factor": 1.0, "high_freq_factor": 4.0,
                              "original_max_position_embeddings": 8192}),
    "yarn_x4": dict(theta=1e4, dim=64, max_seq=32, mpe=32,
                    scaling={"rope_type": "yarn", "factor": 4.0, "original_max_position_embeddings": 32}),
}
out = {}
for name, c in CASES.items():
    cos, sin = O.rope_tables_scaled(c["theta"], c["dim"], c["max_seq"], c["scaling"], c["mpe"])
    rows = [1, 7, cos.shape[0] - 1]
    out[name] = {"args": {k: v for k, v in c.items()}, "n": int(cos.shape[0]), "rows": rows,
                 "cos": [[float(x) for x in cos[r]] for r in rows], "sin": [[float(x) for x in sin[r]] for r in rows]}
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "rope_tables.json"), "w"), indent=0)
print("wrote", list(out))
