"""Third-party anchor for the RoPE scaling builders: inverse frequencies and attention factor from Hugging Face
transformers' `ROPE_INIT_FUNCTIONS` (the published implementations the reference's `ScalingRotaryEmbedding::new`
follows, src/openai/models/layers/rotary_emb.rs:107-341) -> tests/golden/rope_hf.json.

Run here (transformers is importable in the build container):  python tests/golden/make_golden_rope_hf.py
`dynamic` is left out on purpose: the reference raises (theta * bracket) to dim/(dim-2) (rotary_emb.rs:240-252) where
transformers computes theta * bracket**(dim/(dim-2)); the oracle follows the reference."""
import json
import os

import torch  # noqa: F401
import transformers
from transformers import LlamaConfig
from transformers.modeling_rope_utils import ROPE_INIT_FUNCTIONS

CASES = {
    "llama3.1": dict(theta=5e5, dim=128, mpe=131072, max_seq=700,
                     scaling={"rope_type": "llama3", "factor": 8.0, "low_freq_factor": 1.0, "high_freq_factor": 4.0,
                              "original_max_position_embeddings": 8192}),
    "llama3.2-1b-shape": dict(theta=5e5, dim=64, mpe=131072, max_seq=512,
                              scaling={"rope_type": "llama3", "factor": 32.0, "low_freq_factor": 1.0,
                                       "high_freq_factor": 4.0, "original_max_position_embeddings": 8192}),
    "linear": dict(theta=1e4, dim=64, mpe=512, max_seq=300, scaling={"rope_type": "linear", "factor": 4.0}),
    "yarn": dict(theta=1e4, dim=128, mpe=256, max_seq=256,
                 scaling={"rope_type": "yarn", "factor": 4.0, "original_max_position_embeddings": 256, "beta_fast": 32.0,
                          "beta_slow": 1.0}),
    "yarn-qwen-shape": dict(theta=1e6, dim=128, mpe=32768, max_seq=1024,
                            scaling={"rope_type": "yarn", "factor": 4.0, "original_max_position_embeddings": 32768}),
}


def main():
    out = {"_made_with": "transformers " + transformers.__version__}
    for name, c in CASES.items():
        rp = dict(c["scaling"])
        rp["rope_theta"] = c["theta"]
        heads = 4
        cfg = LlamaConfig(hidden_size=c["dim"] * heads, num_attention_heads=heads, head_dim=c["dim"],
                          max_position_embeddings=c["mpe"], rope_parameters=rp)
        inv, att = ROPE_INIT_FUNCTIONS[rp["rope_type"]](cfg, "cpu")
        out[name] = {"args": c, "inv_freq": [float(v) for v in inv.double()], "attention_factor": float(att)}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rope_hf.json")
    json.dump(out, open(path, "w"), indent=0)
    print("wrote", path)


if __name__ == "__main__":
    main()
