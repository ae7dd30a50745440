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
        L.orc_num_threads.restype = i32
        L.orc_set_num_threads.argtypes = [i32]
        _lib = L
    return _lib


def qmatmul(x, blocks, ggml_type, o2):
    x = np.ascontiguousarray(x, np.float32)
    b = np.ascontiguousarray(blocks, np.uint8)
    N, K, T = b.shape[0], b.shape[1] * 256, x.shape[0]
    y = np.empty((T, N), np.float32)
    lib().orc_qmatmul(b.ctypes.data, ggml_type, N, K, x.ctypes.data, T, y.ctypes.data, 1 if o2 else 0)
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

    def __init__(self, cfg, W=None, types=None, seed=1):
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
            if lib().orc_llama_fill_random(self.h, t.ctypes.data, seed) != 0:
                raise MemoryError("oracle weights")

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_llama_destroy(self.h)
            self.h = None

    def decode(self, meta, kv_caches, o2=False):
        """kv_caches: list of (K,V) uint16 arrays in FLASH layout (modified in place)."""
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
                               ctx.ctypes.data, B, bt.shape[1], kp, vp, logits.ctypes.data, 1 if o2 else 0)
        return logits
