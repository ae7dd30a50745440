"""RoPE table builders (host C++ behind the C ABI) vs the numpy restatement of the reference's
ScalingRotaryEmbedding::new (src/openai/models/layers/rotary_emb.rs:107-341,358-457).  No GPU."""
import ctypes

import numpy as np
import pytest

from oracle import ops as O

CASES = [
    ("default", None, dict(theta=5e5, dim=128, max_seq=512, mpe=0)),
    ("llama3.1", {"rope_type": "llama3", "factor": 8.0, "low_freq_factor": 1.0, "high_freq_factor": 4.0,
                  "original_max_position_embeddings": 8192}, dict(theta=5e5, dim=128, max_seq=700, mpe=131072)),
    ("linear", {"type": "linear", "factor": 4.0}, dict(theta=1e4, dim=64, max_seq=300, mpe=512)),
    ("dynamic-alpha", {"rope_type": "dynamic", "alpha": 2.5}, dict(theta=1e4, dim=128, max_seq=256, mpe=384)),
    ("dynamic-factor", {"rope_type": "dynamic", "factor": 2.0, "original_max_position_embeddings": 128},
     dict(theta=1e4, dim=96, max_seq=256, mpe=256)),
    ("yarn", {"rope_type": "yarn", "factor": 4.0, "original_max_position_embeddings": 256, "beta_fast": 32.0,
              "beta_slow": 1.0}, dict(theta=1e4, dim=128, max_seq=256, mpe=256)),
    ("partial-rotary", None, dict(theta=1e4, dim=20, max_seq=128, mpe=0)),
]
TYPES = {"default": 0, "linear": 1, "llama3": 2, "dynamic": 3, "yarn": 4}


@pytest.mark.parametrize("name,scaling,kw", CASES, ids=[c[0] for c in CASES])
def test_rope_tables_match_the_reference_restatement(name, scaling, kw):
    import __graft_entry__ as ge
    ge.build()
    from candle_vllm_amd._lib import lib, RopeScaling
    ref_c, ref_s = O.rope_tables_scaled(kw["theta"], kw["dim"], kw["max_seq"], scaling, kw["mpe"])
    sc = None
    if scaling is not None:
        sc = RopeScaling()
        sc.type = TYPES[scaling.get("rope_type", scaling.get("type"))]
        for k in ("factor", "low_freq_factor", "high_freq_factor", "original_max_position_embeddings", "alpha",
                  "beta_fast", "beta_slow", "attn_factor", "extrapolation_factor"):
            setattr(sc, k, float(scaling.get(k, 0.0)))
    scp = ctypes.byref(sc) if sc is not None else None
    n = lib.mi355_rope_table_len(scp, kw["max_seq"], kw["mpe"])
    assert n == ref_c.shape[0], (n, ref_c.shape)
    cos = np.empty((n, kw["dim"] // 2), np.float32)
    sin = np.empty_like(cos)
    assert lib.mi355_rope_tables(cos.ctypes.data, sin.ctypes.data, kw["dim"], n, kw["theta"], scp, kw["max_seq"], kw["mpe"]) == 0
    # same formulas in f32; powf / cosf of two math libraries may differ by 1 ulp, and 1 ulp of an inverse frequency is
    # an angle error of position * 2^-24 at the highest frequencies
    tol = max(4e-7, 1.2e-7 * n) * max(1.0, float(np.abs(ref_c).max()))
    assert np.abs(cos - ref_c).max() <= tol and np.abs(sin - ref_s).max() <= tol, (np.abs(cos - ref_c).max(), np.abs(sin - ref_s).max())
    if name == "llama3.1":                                            # the scaling actually changed the low frequencies
        d_c, _ = O.rope_tables(kw["theta"], kw["dim"], kw["max_seq"])
        assert np.abs(d_c - ref_c).max() > 1e-2 and np.array_equal(d_c[:, :8], ref_c[:, :8])


def test_bad_arguments_are_rejected():
    import __graft_entry__ as ge
    ge.build()
    from candle_vllm_amd._lib import lib, RopeScaling
    buf = np.empty((4, 4), np.float32)
    sc = RopeScaling()
    sc.type = 2                                                       # llama3 without its factors
    assert lib.mi355_rope_tables(buf.ctypes.data, buf.ctypes.data, 8, 4, 1e4, ctypes.byref(sc), 4, 0) != 0
    assert lib.mi355_rope_tables(buf.ctypes.data, buf.ctypes.data, 7, 4, 1e4, None, 4, 0) != 0      # odd rotary dim
    sc.type = 9
    assert lib.mi355_rope_table_len(ctypes.byref(sc), 4, 0) == -1


def test_rope_tables_match_committed_golden_rows():
    """fixture made by tests/golden/make_golden_rope.py: pins the oracle AND the library's builder"""
    import json
    import os
    import __graft_entry__ as ge
    ge.build()
    from candle_vllm_amd._lib import lib, RopeScaling
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "rope_tables.json")))
    for name, g in gold.items():
        a = g["args"]
        oc, osn = O.rope_tables_scaled(a["theta"], a["dim"], a["max_seq"], a["scaling"], a["mpe"])
        sc = None
        if a["scaling"] is not None:
            sc = RopeScaling()
            sc.type = TYPES[a["scaling"]["rope_type"]]
            for k in ("factor", "low_freq_factor", "high_freq_factor", "original_max_position_embeddings", "alpha",
                      "beta_fast", "beta_slow", "attn_factor", "extrapolation_factor"):
                setattr(sc, k, float(a["scaling"].get(k, 0.0)))
        scp = ctypes.byref(sc) if sc is not None else None
        n = lib.mi355_rope_table_len(scp, a["max_seq"], a["mpe"])
        assert n == g["n"] == oc.shape[0]
        cos = np.empty((n, a["dim"] // 2), np.float32)
        sin = np.empty_like(cos)
        assert lib.mi355_rope_tables(cos.ctypes.data, sin.ctypes.data, a["dim"], n, a["theta"], scp, a["max_seq"], a["mpe"]) == 0
        for r, gc, gs in zip(g["rows"], g["cos"], g["sin"]):
            tol = max(4e-7, 1.2e-7 * n) * max(1.0, float(np.abs(np.asarray(gc)).max()))
            assert np.abs(oc[r] - np.asarray(gc, np.float32)).max() <= 1e-7 and np.abs(osn[r] - np.asarray(gs, np.float32)).max() <= 1e-7
            assert np.abs(cos[r] - np.asarray(gc, np.float32)).max() <= tol and np.abs(sin[r] - np.asarray(gs, np.float32)).max() <= tol


def test_rope_scaling_matches_huggingface_inverse_frequencies():
    """third-party anchor (tests/golden/rope_hf.json, made by tests/golden/make_golden_rope_hf.py from transformers'
    ROPE_INIT_FUNCTIONS): the oracle's restatement of the reference AND the library's builder must reproduce
    cos/sin(position * inv_freq_HF) * attention_factor_HF for llama3 / linear / yarn.  Tolerance: the angle is formed in
    f32 (position * inv_freq), so an entry may be off by the angle's rounding, angle * 2^-22, plus table rounding."""
    import json
    import os
    import __graft_entry__ as ge
    ge.build()
    from candle_vllm_amd._lib import lib, RopeScaling
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "rope_hf.json")))
    for name, g in gold.items():
        if name.startswith("_"):
            continue
        a = g["args"]
        inv, att = np.asarray(g["inv_freq"], np.float64), float(g["attention_factor"])
        oc, osn = O.rope_tables_scaled(a["theta"], a["dim"], a["max_seq"], a["scaling"], a["mpe"])
        sc = RopeScaling()
        sc.type = TYPES[a["scaling"]["rope_type"]]
        for k in ("factor", "low_freq_factor", "high_freq_factor", "original_max_position_embeddings", "alpha",
                  "beta_fast", "beta_slow", "attn_factor", "extrapolation_factor"):
            setattr(sc, k, float(a["scaling"].get(k, 0.0)))
        n = lib.mi355_rope_table_len(ctypes.byref(sc), a["max_seq"], a["mpe"])
        assert n == oc.shape[0] and inv.shape[0] == a["dim"] // 2
        cos = np.empty((n, a["dim"] // 2), np.float32)
        sin = np.empty_like(cos)
        assert lib.mi355_rope_tables(cos.ctypes.data, sin.ctypes.data, a["dim"], n, a["theta"], ctypes.byref(sc),
                                     a["max_seq"], a["mpe"]) == 0
        rows = sorted({0, 1, 17, a["max_seq"] // 2, a["max_seq"] - 1, n - 1})
        for r in rows:
            ang = r * inv
            tol = att * (ang * 2.0 ** -22 + 2e-6)
            for got_c, got_s, who in ((oc[r], osn[r], "oracle"), (cos[r], sin[r], "library")):
                assert (np.abs(got_c - np.cos(ang) * att) <= tol).all(), (name, who, r, np.abs(got_c - np.cos(ang) * att).max())
                assert (np.abs(got_s - np.sin(ang) * att) <= tol).all(), (name, who, r, np.abs(got_s - np.sin(ang) * att).max())


def test_effective_max_seq_len_matches_the_reference_unit_tests():
    """`Config::effective_max_seq_len` / `apply_runtime_rope_overrides` against the reference's own tests
    (src/openai/models/mod.rs:889-918): yarn factor 4 over 262144 -> 1048576; runtime yarn factor 8 -> 2097152"""
    import __graft_entry__ as ge
    ge.build()
    from candle_vllm_amd._lib import lib, RopeScaling
    sc = RopeScaling()
    sc.type, sc.factor, sc.original_max_position_embeddings = TYPES["yarn"], 4.0, 262144.0
    assert lib.mi355_effective_max_seq_len(ctypes.addressof(sc), 262144) == 1_048_576      # test_effective_max_seq_len_scales_yarn_context
    sc.factor = 8.0                                                                        # apply_runtime_rope_overrides(Some(8.0))
    assert lib.mi355_effective_max_seq_len(ctypes.addressof(sc), 262144) == 2_097_152
    # and the rules around them (:691-701): only yarn, only factor > 1, never below the base, absent original -> base
    assert lib.mi355_effective_max_seq_len(None, 8192) == 8192
    sc.factor = 1.0
    assert lib.mi355_effective_max_seq_len(ctypes.addressof(sc), 262144) == 262144
    sc.factor, sc.original_max_position_embeddings = 4.0, 1024.0
    assert lib.mi355_effective_max_seq_len(ctypes.addressof(sc), 8192) == 8192
    sc.original_max_position_embeddings = 0.0
    assert lib.mi355_effective_max_seq_len(ctypes.addressof(sc), 8192) == 8192
    sc.type, sc.factor, sc.original_max_position_embeddings = TYPES["llama3"], 8.0, 8192.0
    assert lib.mi355_effective_max_seq_len(ctypes.addressof(sc), 131072) == 131072
    sc.type, sc.factor, sc.original_max_position_embeddings = TYPES["yarn"], 2.5, 1001.0
    assert lib.mi355_effective_max_seq_len(ctypes.addressof(sc), 100) == 2503                # round(2502.5) away from zero
