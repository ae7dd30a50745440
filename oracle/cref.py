"""ctypes binding of the C twin of the oracle (oracle/oracle.c -> oracle/liboracle.so).
TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")


def build():
    import subprocess
    src = os.path.join(_HERE, "oracle.c")
    if not os.path.exists(LIB_PATH) or os.path.getmtime(src) > os.path.getmtime(LIB_PATH):
        subprocess.check_call(["gcc", "-O3", "-march=x86-64-v3", "-fopenmp", "-shared", "-fPIC", "-o", LIB_PATH, src, "-lm"])


class Cfg(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("hidden", "n_layers", "n_heads", "n_kv_heads", "head_dim",
                                              "intermediate", "vocab", "max_seq", "block_size")] + \
               [("rms_eps", ctypes.c_float), ("rope_theta", ctypes.c_float)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = ctypes.CDLL(LIB_PATH)
        vp, i32, i64, f32 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_float
        L.orc_qmatmul.argtypes = [vp, i32, i32, i32, vp, i32, vp, i32]
        L.orc_dequantize_q4k.argtypes = [vp, vp, i64]
        L.orc_dequantize_q6k.argtypes = [vp, vp, i64]
        L.orc_llama_create.restype = vp
        L.orc_llama_create.argtypes = [ctypes.POINTER(Cfg)]
        L.orc_llama_destroy.argtypes = [vp]
        L.orc_llama_set_qweight.argtypes = [vp, i32, i32, i32, vp, i32, i32]
        L.orc_llama_set_f32.argtypes = [vp, i32, i32, vp]
        L.orc_llama_fill_random.argtypes = [vp, vp, ctypes.c_uint64]
        L.orc_llama_fill_random.restype = i32
        L.orc_llama_decode.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, vp, vp, vp, i32]
        L.orc_llama_get_qweight.restype = vp
        L.orc_llama_get_qweight.argtypes = [vp, i32, i32, ctypes.POINTER(i32), ctypes.POINTER(i32), ctypes.POINTER(i32)]
        L.orc_llama_set_trace.argtypes = [vp, vp]
        L.orc_llama_set_fill_scale.argtypes = [f32]
        L.orc_llama_set_attn_bf16.argtypes = [i32]
        L.orc_llama_set_trace_parts.argtypes = [vp, vp, vp, vp, vp]
        L.orc_llama_prefill.argtypes = [vp, vp, vp, vp, i32, vp, vp, vp]
        L.orc_bf16_gemv.argtypes = [vp, i32, i32, vp, vp]
        L.orc_num_threads.restype = i32
        L.orc_set_num_threads.argtypes = [i32]
        L.orc_llama_phase_times.argtypes = [vp, i32]
        _lib = L
    return _lib


def qmatmul(x, blocks, ggml_type, o2):
    x = np.ascontiguousarray(x, np.float32)
    b = np.ascontiguousarray(blocks, np.uint8)
    N, K, T = b.shape[0], b.shape[1] * 256, x.shape[0]
    y = np.empty((T, N), np.float32)
    lib().orc_qmatmul(b.ctypes.data, ggml_type, N, K, x.ctypes.data, T, y.ctypes.data, int(o2))
    return y


def bf16_gemv(w_bits, x):
    """y = W . x with W [N,K] bf16 bit patterns (uint16), x f32 [K]; f32 accumulation (AVX2 + OpenMP)"""
    w = np.ascontiguousarray(w_bits, np.uint16)
    x = np.ascontiguousarray(x, np.float32)
    y = np.empty(w.shape[0], np.float32)
    lib().orc_bf16_gemv(w.ctypes.data, w.shape[0], w.shape[1], x.ctypes.data, y.ctypes.data)
    return y


def make_cfg(cfg):
    c = Cfg()
    for n in ("hidden", "n_layers", "n_heads", "n_kv_heads", "head_dim", "intermediate", "vocab", "max_seq",
              "block_size"):
        setattr(c, n, getattr(cfg, n))
    c.rms_eps, c.rope_theta = cfg.rms_eps, cfg.rope_theta
    return c


class CLlama:
    """C decode step over numpy-owned weights (oracle.llama.make_weights dict) or random weights."""
    _SLOT = {"wq": 0, "wk": 1, "wv": 2, "wo": 3, "w1": 4, "w2": 5, "w3": 6}

    def __init__(self, cfg, W=None, types=None, seed=1, fill_scale=1.0):
        """fill_scale (random weights only): multiplies the super-block scales, i.e. the std of the dequantised weights
        (1.0 = std ~0.04, the bench's synthetic weights; 0.2 = branch gain < 1 as in a trained checkpoint)"""
        self.cfg = cfg
        self.h = lib().orc_llama_create(ctypes.byref(make_cfg(cfg)))
        self._keep = []
        if W is not None:
            def f(a):
                a = np.ascontiguousarray(a, np.float32)
                self._keep.append(a)
                return a.ctypes.data

            def q(layer, which, tw):
                t, blocks = tw
                b = np.ascontiguousarray(blocks, np.uint8)
                self._keep.append(b)
                lib().orc_llama_set_qweight(self.h, layer, which, t, b.ctypes.data, b.shape[0], b.shape[1] * 256)
            lib().orc_llama_set_f32(self.h, -1, 9, f(W["tok_embd"]))
            lib().orc_llama_set_f32(self.h, -1, 10, f(W["output_norm"]))
            q(-1, 11, W["output"])
            for l, lw in enumerate(W["layers"]):
                lib().orc_llama_set_f32(self.h, l, 7, f(lw["attn_norm"]))
                lib().orc_llama_set_f32(self.h, l, 8, f(lw["ffn_norm"]))
                for name, slot in self._SLOT.items():
                    q(l, slot, lw[name])
        else:
            t = np.ascontiguousarray(types, np.int32)
            lib().orc_llama_set_fill_scale(float(fill_scale))
            rc = lib().orc_llama_fill_random(self.h, t.ctypes.data, seed)
            lib().orc_llama_set_fill_scale(1.0)
            if rc != 0:
                raise MemoryError("oracle weights")

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_llama_destroy(self.h)
            self.h = None

    def set_f32(self, layer, which, a):
        """which: 7 attn_norm, 8 ffn_norm; layer -1: 9 tok_embd [vocab, hidden], 10 output_norm (borrowed: kept alive here)"""
        a = np.ascontiguousarray(a, np.float32)
        self._keep.append(a)
        lib().orc_llama_set_f32(self.h, layer, which, a.ctypes.data)

    def qweight(self, layer, which):
        """(pointer, ggml type, rows, cols) of one tensor's native GGUF blocks (layer -1: the output matrix)"""
        t, n, k = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
        p = lib().orc_llama_get_qweight(self.h, layer, which, ctypes.byref(t), ctypes.byref(n), ctypes.byref(k))
        return p, t.value, n.value, k.value

    def set_trace(self, trace):
        """trace: f32 [n_layers+1, B, hidden] filled by the next decode() calls (None = off); kept alive here"""
        self._trace = trace
        lib().orc_llama_set_trace(self.h, None if trace is None else trace.ctypes.data)

    def set_trace_parts(self, q=None, att=None, mid=None, h=None):
        """f32 [L,B,H*D], [L,B,H*D], [L,B,hidden], [L,B,I] filled by the next decode() calls (all None = off)"""
        self._trace_parts = (q, att, mid, h)
        lib().orc_llama_set_trace_parts(self.h, *[None if a is None else a.ctypes.data for a in (q, att, mid, h)])

    def prefill(self, tokens, positions, slots, kv_caches):
        """one prompt step of ONE sequence without a cached prefix (O1f products); returns the last token's logits [vocab]"""
        tok = np.ascontiguousarray(tokens, np.uint32)
        pos = np.ascontiguousarray(positions, np.int64)
        sl = np.ascontiguousarray(slots, np.int64)
        L = len(kv_caches)
        kp = (ctypes.c_void_p * L)(*[k.ctypes.data for k, _ in kv_caches])
        vp = (ctypes.c_void_p * L)(*[v.ctypes.data for _, v in kv_caches])
        logits = np.empty(self.cfg.vocab, np.float32)
        lib().orc_llama_prefill(self.h, tok.ctypes.data, pos.ctypes.data, sl.ctypes.data, len(tok), kp, vp, logits.ctypes.data)
        return logits

    def decode(self, meta, kv_caches, o2=False):
        """kv_caches: list of (K,V) uint16 arrays in FLASH layout (modified in place).  o2: False/0 = O1 (f64 dots),
        True/1 = candle-CPU Q8_K integer dots, 2 = O1f (f32 blocked dots, for many-token steps)."""
        B = len(meta["input_ids"])
        tok = np.ascontiguousarray(meta["input_ids"], np.uint32)
        pos = np.ascontiguousarray(meta["positions"], np.int64)
        slots = np.ascontiguousarray(meta["slot_mapping"], np.int64)
        bt = np.ascontiguousarray(meta["block_tables"], np.uint32)
        ctx = np.ascontiguousarray(meta["context_lens"], np.uint32)
        L = len(kv_caches)
        kp = (ctypes.c_void_p * L)(*[k.ctypes.data for k, _ in kv_caches])
        vp = (ctypes.c_void_p * L)(*[v.ctypes.data for _, v in kv_caches])
        logits = np.empty((B, self.cfg.vocab), np.float32)
        lib().orc_llama_decode(self.h, tok.ctypes.data, pos.ctypes.data, slots.ctypes.data, bt.ctypes.data,
                               ctx.ctypes.data, B, bt.shape[1], kp, vp, logits.ctypes.data, int(o2))
        return logits
