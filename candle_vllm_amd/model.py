"""GGUFLLaMa on the MI355X decode path: thin ctypes wrapper over the C++ host layer (csrc/host_model.cpp),
which mirrors src/openai/models/quantized_llama.rs (forward), src/scheduler/cache_engine.rs (KV cache) and
src/backend/graph.rs (decode graph).  torch only supplies streams / device memory for tests and the bench."""
import ctypes
import os

import numpy as np
import torch

from ._lib import lib, LlamaConfig
from .ops import _check, GGML_Q4_K, GGML_Q6_K, KV_FLASH, KV_PAGED, DT_F32, DT_BF16  # noqa: F401

KV_PAGED_FP8 = 2          # `--kvcache-dtype fp8`: e4m3fn bytes in the paged layout with x = 16

W_WQ, W_WK, W_WV, W_WO, W_W1, W_W2, W_W3, W_ATTN_NORM, W_FFN_NORM, W_TOK_EMBD, W_OUTPUT_NORM, W_OUTPUT = range(12)
_SLOT = {"wq": W_WQ, "wk": W_WK, "wv": W_WV, "wo": W_WO, "w1": W_W1, "w2": W_W2, "w3": W_W3}
_TILE_BYTES = {GGML_Q4_K: 2304, GGML_Q6_K: 3360}


class ModelDims:
    """dims a GGUFLLaMa is created from (read from the GGUF metadata at quantized_llama.rs:231-260)"""

    def __init__(self, hidden=4096, n_layers=32, n_heads=32, n_kv_heads=8, head_dim=128, intermediate=14336, vocab=128256,
                 rms_eps=1e-5, rope_theta=500000.0, max_seq=8192, block_size=64):
        self.hidden, self.n_layers, self.n_heads, self.n_kv_heads, self.head_dim = hidden, n_layers, n_heads, n_kv_heads, head_dim
        self.intermediate, self.vocab, self.rms_eps, self.rope_theta = intermediate, vocab, rms_eps, rope_theta
        self.max_seq, self.block_size = max_seq, block_size

    @staticmethod
    def llama3_8b():
        """Llama-3-8B (public model-card values): BASELINE.json configs[1]"""
        return ModelDims()


def q4km_type_for(name, layer, n_layers):
    """llama.cpp Q4_K_M mixture [EXT]: output Q6_K; attn_v / ffn_down Q6_K on `use_more_bits` layers."""
    if name == "output":
        return GGML_Q6_K
    if name in ("wv", "w2"):
        n8 = n_layers // 8
        if layer < n8 or layer >= 7 * n8 or (layer - n8) % 3 == 2:
            return GGML_Q6_K
    return GGML_Q4_K


def random_tiles(ggml_type, n_rows, k, device, gen, d_scale=0.0025):
    """Random but VALID repacked weights made directly on the GPU: random codes and sub-block scales,
    f16 super-block scales chosen so that dequantised weights have std ~0.02 (synthetic llama weights)."""
    nb = lib.mi355_qweight_repacked_size(ggml_type, n_rows, k)
    t = torch.randint(0, 256, (nb,), dtype=torch.uint8, device=device, generator=gen)
    ntile = (n_rows + 15) // 16 * (k // 256)
    if ggml_type == GGML_Q4_K:
        # w = d*sc*q - dmin*m with sc,m ~U[0,63], q ~U[0,15]: d = dmin*2.3 centres the weights near zero
        d = np.array([d_scale * 0.08], np.float16).view(np.uint8)
        dm = np.array([d_scale * 0.6], np.float16).view(np.uint8)
        tv = t.view(ntile, 2304)
        tv[:, 0:256:16] = int(d[0]); tv[:, 1:256:16] = int(d[1])
        tv[:, 2:256:16] = int(dm[0]); tv[:, 3:256:16] = int(dm[1])
    else:
        d = np.array([d_scale * 0.005], np.float16).view(np.uint8)
        tv = t.view(ntile, 3360)
        tv[:, 3328:3360:2] = int(d[0]); tv[:, 3329:3360:2] = int(d[1])
    return t


# ---- control-plane helpers of the tensor-parallel set-up (shared by GGUFLLaMa.init_comm and bench.py's fallback path) ---------------
def _ctl_device(dist):
    """where the launcher's small agreement tensors live: on the device under the nccl (RCCL) backend, on the host under gloo"""
    get = getattr(dist, "get_backend", None)            # (single-process callers pass a stub with `broadcast` only)
    return "cuda" if get is None or get() == "nccl" else "cpu"


def comm_all_min(dist, v, group=None):
    """the ranks decide together: minimum of an int flag"""
    f = torch.tensor([int(v)], dtype=torch.int32, device=_ctl_device(dist) if group is None else "cpu")
    dist.all_reduce(f, op=dist.ReduceOp.MIN, group=group)
    return int(f.item())


def ranks_have_peer_access(dist, world, group=None):
    """local answer: can THIS rank's device reach the device of every other rank?  Only the devices the ranks actually use are asked
    (an unrelated GPU without peer access must not veto the peer kernel); ranks that cannot tell (one visible device per process, e.g.
    HIP_VISIBLE_DEVICES) answer yes and leave the decision to export / attach / self-test."""
    try:
        dev = torch.cuda.current_device()
        mine = (os.environ.get("HIP_VISIBLE_DEVICES", os.environ.get("CUDA_VISIBLE_DEVICES", "")), dev)
        used = [None] * world
        dist.all_gather_object(used, mine, group=group)
        if torch.cuda.device_count() < 2 or any(u[0] != mine[0] for u in used):
            return True
        return all(torch.cuda.can_device_access_peer(dev, u[1]) for u in used if u[1] != dev)
    except Exception:
        return False


def p2p_self_test(comm, rank, world):
    """local answer: known sums through the one-shot peer kernel -- two f32 rounds and a bf16 round (the wire dtype of the 16-bit path
    and of wire_bf16), result and sticky error word checked.  Every rank must call it (the kernel waits for its peers)."""
    st = torch.cuda.current_stream().cuda_stream
    ok = True
    for rnd, want, dt, tdt in ((1, world * (world + 1) // 2, DT_F32, torch.float32), (2, world * world, DT_F32, torch.float32),
                               (1, world * (world + 1) // 2, DT_BF16, torch.bfloat16)):
        x = torch.full((2048,), float(rnd * rank + 1), dtype=tdt, device="cuda")
        rc = lib.mi355_comm_all_reduce(comm, x.data_ptr(), x.numel(), dt, st)
        torch.cuda.synchronize()
        ok = ok and rc == 0 and bool((x.float() == float(want)).all().item()) and lib.mi355_comm_p2p_error(comm) == 0
    return ok


def comm_attach_p2p(dist, comm, rank, world, group=None):
    """export -> gather the 64-byte IPC handles -> attach -> self-test, the same collective sequence on every rank whatever fails locally;
    -> True when the peer kernel is usable on EVERY rank"""
    h = ctypes.create_string_buffer(64)
    exported = int(lib.mi355_comm_p2p_export(comm, ctypes.addressof(h)) == 0)
    all_h = [None] * world
    dist.all_gather_object(all_h, bytes(h.raw), group=group)
    attached = 0
    if comm_all_min(dist, exported, group):
        blob = ctypes.create_string_buffer(b"".join(all_h), 64 * world)
        attached = int(lib.mi355_comm_p2p_attach(comm, ctypes.addressof(blob), rank, world) == 0)
    ok = comm_all_min(dist, attached, group)
    if ok:
        try:
            ok = p2p_self_test(comm, rank, world)
        except Exception:
            ok = False
        ok = comm_all_min(dist, ok, group)
    return bool(ok)


class GGUFLLaMa:
    def __init__(self, cfg, max_batch=1, max_blocks_per_seq=None, kv_layout=KV_FLASH, tp_rank=0, tp_world=1):
        """cfg: oracle.llama.LlamaConfig-like (hidden, n_layers, n_heads, n_kv_heads, head_dim, intermediate, vocab,
        rms_eps, rope_theta, max_seq, block_size)."""
        if not torch.cuda.is_available():
            raise RuntimeError("GGUFLLaMa needs the MI355X: there is no CPU fallback")
        self.cfg = cfg
        c = LlamaConfig()
        c.hidden, c.n_layers, c.n_heads, c.n_kv_heads = cfg.hidden, cfg.n_layers, cfg.n_heads, cfg.n_kv_heads
        c.head_dim, c.intermediate, c.vocab = cfg.head_dim, cfg.intermediate, cfg.vocab
        c.max_seq, c.block_size, c.kv_layout, c.max_batch = cfg.max_seq, cfg.block_size, kv_layout, max_batch
        c.max_blocks_per_seq = max_blocks_per_seq or -(-cfg.max_seq // cfg.block_size)
        c.rms_eps, c.rope_theta, c.tp_rank, c.tp_world = cfg.rms_eps, cfg.rope_theta, tp_rank, tp_world
        c.n_expert = int(getattr(cfg, "n_expert", 0) or 0)            # Mixtral-style MoE MLP (llama.expert_count)
        c.n_expert_used = int(getattr(cfg, "n_expert_used", 0) or 0)
        self.c = c
        self.kv_layout = kv_layout
        self.max_batch = max_batch
        self.h = lib.mi355_llama_create(ctypes.byref(c))
        if not self.h:
            raise RuntimeError("mi355_llama_create failed")
        self._keep = []
        self.weight_bytes = 0            # local (this rank's) quantised weight bytes touched per step
        self.tp_rank, self.tp_world = tp_rank, tp_world
        self.part_bytes = {}             # (layer, part) -> algorithmic weight bytes of that launch group
        if cfg.n_heads % tp_world:
            raise ValueError("num_attention_heads must be divisible by the TP world size (attention.rs:553-554)")
        self.local_heads = cfg.n_heads // tp_world
        self.local_kv_heads = max(cfg.n_kv_heads // tp_world, 1)

    def __del__(self):
        if getattr(self, "h", None) and lib is not None:           # `lib` is already gone at interpreter shutdown
            lib.mi355_llama_destroy(self.h)
            self.h = None

    @classmethod
    def from_gguf(cls, path, max_batch=1, max_blocks_per_seq=64, block_size=64, kv_layout=KV_FLASH, max_seq=0,
                  tp_rank=0, tp_world=1):
        """`GGUFLLaMa::from_gguf` (quantized_llama.rs:203-420): config from the file's metadata, all tensors loaded
        by the C++ reader (mi355_llama_load_gguf); with tp_world > 1 every rank keeps its raw byte-range shard
        (mi355_llama_load_gguf_tp) and needs `init_comm` / `set_comm` before the first step."""
        if not torch.cuda.is_available():
            raise RuntimeError("GGUFLLaMa needs the MI355X: there is no CPU fallback")
        h = ctypes.c_void_p(0)
        c = LlamaConfig()
        _check(lib.mi355_llama_load_gguf_tp(str(path).encode(), max_batch, max_blocks_per_seq, block_size, kv_layout,
                                            max_seq, tp_rank, tp_world, ctypes.addressof(h), ctypes.addressof(c)),
               "load_gguf")
        self = cls.__new__(cls)
        from types import SimpleNamespace
        self.cfg = SimpleNamespace(hidden=c.hidden, n_layers=c.n_layers, n_heads=c.n_heads, n_kv_heads=c.n_kv_heads,
                                   head_dim=c.head_dim, intermediate=c.intermediate, vocab=c.vocab, max_seq=c.max_seq,
                                   block_size=c.block_size, rms_eps=c.rms_eps, rope_theta=c.rope_theta)
        self.c, self.h, self.kv_layout, self.max_batch = c, h.value, kv_layout, max_batch
        self._keep, self.weight_bytes, self.part_bytes = [], 0, {}
        self.tp_rank, self.tp_world = tp_rank, tp_world
        self.local_heads, self.local_kv_heads = c.n_heads // tp_world, max(c.n_kv_heads // tp_world, 1)
        return self

    # ------------------------------------------------------------------ weights
    def load_oracle_weights(self, W):
        """W: dict produced by oracle.llama.make_weights (native GGUF blocks on the host)."""
        def f32(layer, which, a):
            a = np.ascontiguousarray(a, np.float32)
            _check(lib.mi355_llama_set_f32(self.h, layer, which, a.ctypes.data, a.size), "set_f32")

        def qw(layer, which, tw):
            t, blocks = tw
            b = np.ascontiguousarray(blocks)
            # blocks [N, k / per, block bytes]: a k-quant super-block holds 256 weights, a re-quantised Q8_0 shard's block 32
            # (tp.shard_weights(..., requant); ADVICE r2: `* 256` for every type read 8x the host buffer)
            per = 32 if int(t) == 8 else 256
            _check(lib.mi355_llama_set_qweight(self.h, layer, which, t, b.ctypes.data, b.shape[0], b.shape[1] * per),
                   "set_qweight")
            self.weight_bytes += b.size
        f32(-1, W_TOK_EMBD, W["tok_embd"])
        f32(-1, W_OUTPUT_NORM, W["output_norm"])
        qw(-1, W_OUTPUT, W["output"])
        for l, lw in enumerate(W["layers"]):
            f32(l, W_ATTN_NORM, lw["attn_norm"])
            f32(l, W_FFN_NORM, lw["ffn_norm"])
            if "experts" in lw:                                        # MoE layer (quantized_llama.rs:347-365)
                f32(l, 12, lw["gate_inp"])
                for e, ex in enumerate(lw["experts"]):
                    for name in ("w1", "w2", "w3"):
                        t, blocks = ex[name]
                        b = np.ascontiguousarray(blocks)
                        _check(lib.mi355_llama_set_moe_expert(self.h, l, _SLOT[name], e, t, b.ctypes.data, b.shape[0],
                                                              b.shape[1] * 256), "set_moe_expert")
                        self.weight_bytes += b.size
                for name in ("wq", "wk", "wv", "wo"):
                    qw(l, _SLOT[name], lw[name])
                continue
            for name, slot in _SLOT.items():
                qw(l, slot, lw[name])

    def load_synthetic(self, seed=1235, recipe="q4_k_m", scale=1.0):
        """Random-init weights of the configured architecture, generated on the GPU already in tile order.  scale: std of the
        projections relative to the default (1.0 = std ~0.04: every branch has gain >> 1 and the 32-layer stack amplifies rounding
        noise chaotically -- fine for timing; 0.2 = branch gain < 1 as in a trained checkpoint, for comparisons of logits).
        Tensor parallel (round 6): every rank draws the GLOBAL tensor from the same seed and keeps its shard -- row tiles for the
        column-parallel projections and the lm_head, k-blocks for wo / w2, the kv-head group of `kv_head_shard` for wk / wv
        (distributed.rs:243-249,696-765) -- so the sharded model IS the one-GPU model and their tokens can be compared."""
        from .tp import kv_head_shard
        cfg = self.cfg
        gen = torch.Generator(device="cuda")
        gen.manual_seed(seed)
        Wn, R = self.tp_world, self.tp_rank
        Hg, Hkvg, D, hid = cfg.n_heads, cfg.n_kv_heads, cfg.head_dim, cfg.hidden
        Ig, Vg = cfg.intermediate, cfg.vocab
        H, Hkv = self.local_heads, self.local_kv_heads
        I, V = Ig // Wn, Vg // Wn
        if (I % 256) or (Ig % Wn) or (Vg % (16 * Wn)) or ((H * D) % 256):
            raise ValueError("TP shard is not block aligned (the reference re-quantises to Q8_0 here; not built)")
        _, kv_rank, kv_world = kv_head_shard(Hkvg, R, Wn)
        # name -> (global rows, global k, sharded axis, this rank's index on that axis, shards on that axis)
        plan = {"wq": (Hg * D, hid, 0, R, Wn), "wk": (Hkvg * D, hid, 0, kv_rank, kv_world), "wv": (Hkvg * D, hid, 0, kv_rank, kv_world),
                "wo": (hid, Hg * D, 1, R, Wn), "w1": (Ig, hid, 0, R, Wn), "w3": (Ig, hid, 0, R, Wn), "w2": (hid, Ig, 1, R, Wn),
                "output": (Vg, hid, 0, R, Wn)}
        part_of = {"wq": 0, "wk": 0, "wv": 0, "wo": 2, "w1": 3, "w3": 3, "w2": 4}

        def f32(layer, which, a):
            a = np.ascontiguousarray(a, np.float32)
            _check(lib.mi355_llama_set_f32(self.h, layer, which, a.ctypes.data, a.size), "set_f32")

        def qw(layer, which, name):
            ng, kg, axis, idx, cnt = plan[name]
            t = q4km_type_for(name, layer, cfg.n_layers) if recipe == "q4_k_m" else \
                (GGML_Q6_K if name == "output" else GGML_Q4_K)
            tiles = random_tiles(t, ng, kg, "cuda", gen, d_scale=0.0025 * scale)
            n, k = (ng // cnt, kg) if axis == 0 else (ng, kg // cnt)
            if cnt > 1:                                           # [row tile][k-block][tile bytes]: the shard is a slab of that view
                tv = tiles.view(ng // 16, kg // 256, _TILE_BYTES[t])
                tv = tv[idx * (n // 16):(idx + 1) * (n // 16)] if axis == 0 else tv[:, idx * (k // 256):(idx + 1) * (k // 256)]
                tiles = tv.contiguous().view(-1)
            self._keep.append(tiles)
            _check(lib.mi355_llama_set_qweight_tiles(self.h, layer, which, t, tiles.data_ptr(), n, k), "set_tiles")
            nbytes = (n // 16) * (k // 256) * _TILE_BYTES[t]
            self.weight_bytes += nbytes
            key = (layer, 5 if layer < 0 else part_of[name])
            self.part_bytes[key] = self.part_bytes.get(key, 0) + nbytes
        rng = np.random.default_rng(seed)
        emb = torch.randn((cfg.vocab, hid), device="cuda", generator=gen) * 0.02      # identical on every rank
        f32(-1, W_TOK_EMBD, emb.cpu().numpy())
        del emb
        f32(-1, W_OUTPUT_NORM, 1.0 + rng.normal(0, 0.02, hid))
        qw(-1, W_OUTPUT, "output")
        for l in range(cfg.n_layers):
            f32(l, W_ATTN_NORM, 1.0 + rng.normal(0, 0.02, hid))
            f32(l, W_FFN_NORM, 1.0 + rng.normal(0, 0.02, hid))
            for name, slot in _SLOT.items():
                qw(l, slot, name)
        torch.cuda.synchronize()

    @property
    def weight_bytes_global(self):
        return self.weight_bytes * self.tp_world

    # ------------------------------------------------------------------ tensor parallel
    def init_comm(self, dist, p2p="auto", wire_bf16=False):
        """RCCL communicator for this rank: rank 0 draws the unique id, torch.distributed ships the 128 bytes.
        p2p: the transport of the decode-sized all-reduces (<= 256 KiB; C1 / C2 are 16 KiB at batch 1) --
          False  RCCL on its side stream for every message;
          True   the one-shot peer-to-peer kernel (every rank's 64-byte IPC handle gathered through `dist`), in-stream;
          "auto" (default) the peer kernel when every pair of the devices the ranks actually use has peer access AND a self-test across
                 the ranks (f32 and bf16 all-reduces of known data through the peer kernel, the sticky error word clean) passes on every
                 rank; otherwise every rank goes back to RCCL together.  `self.all_reduce_transport` says what was chosen and why.
        wire_bf16: the reference's all-reduce numerics (attention.rs:1003-1008).
        Every rank runs the SAME sequence of collectives whatever fails locally: a local failure becomes a flag, the ranks take the minimum,
        and all of them raise (or fall back) together -- a rank that raised between two collectives used to leave its peers inside the next
        one."""
        all_min = (lambda v: comm_all_min(dist, v)) if self.tp_world > 1 else (lambda v: int(v))     # a one-rank world agrees with itself
        buf = np.zeros(128, np.uint8)
        rc0 = lib.mi355_comm_unique_id(buf.ctypes.data) if self.tp_rank == 0 else 0
        t = torch.from_numpy(buf).to(_ctl_device(dist))
        dist.broadcast(t, src=0)
        buf = np.ascontiguousarray(t.cpu().numpy())
        if not all_min(rc0 == 0):
            raise RuntimeError("mi355_comm_unique_id failed on rank 0 (librccl not loadable?)")
        rc = lib.mi355_llama_init_comm(self.h, buf.ctypes.data)       # collective inside RCCL: every rank enters it
        if not all_min(rc == 0):
            raise RuntimeError(f"mi355_llama_init_comm failed on at least one rank (here: hipError / ncclResult {rc})")
        comm = lib.mi355_llama_comm_handle(self.h)
        if wire_bf16:
            _check(lib.mi355_comm_set_options(comm, 1, 1), "comm_set_options")
        self.all_reduce_transport = "RCCL on a side stream"
        if not p2p or (p2p == "auto" and self.tp_world < 2):          # (a one-rank world has no peer to reach)
            return self.all_reduce_transport
        if p2p == "auto" and not all_min(ranks_have_peer_access(dist, self.tp_world)):
            self.all_reduce_transport = "RCCL on a side stream (no peer access between every pair of the ranks' devices)"
            return self.all_reduce_transport
        ok = comm_attach_p2p(dist, comm, self.tp_rank, self.tp_world)
        if ok:
            self.all_reduce_transport = "one-shot peer kernel (<= 256 KiB, in-stream), RCCL above"
        else:
            lib.mi355_comm_p2p_enable(comm, 0)
            if p2p is True:
                raise RuntimeError("the one-shot peer all-reduce was requested and failed its export / attach / self-test on at least one rank")
            self.all_reduce_transport = "RCCL on a side stream (the peer kernel failed its export / attach / self-test on at least one rank)"
        return self.all_reduce_transport

    def comm_capture_ok(self, stream):
        """can this stack capture the communicator's all-reduce in a hipGraph?  (local test, nothing goes on the wire)"""
        return lib.mi355_comm_capture_probe(lib.mi355_llama_comm_handle(self.h), stream) == 0

    def set_comm(self, handle):
        """attach a communicator the caller owns (mi355_comm_create / tp.TorchDistComm().handle)"""
        _check(lib.mi355_llama_set_comm(self.h, handle), "set_comm")

    def set_rope_tables(self, cos, sin):
        """replace the default RoPE tables (e.g. llama3 / yarn scaling built by `ops.rope_tables`): f32 [n >= max_seq, D/2]"""
        cos, sin = np.ascontiguousarray(cos, np.float32), np.ascontiguousarray(sin, np.float32)
        _check(lib.mi355_llama_set_rope_tables(self.h, cos.ctypes.data, sin.ctypes.data, cos.shape[0]), "set_rope_tables")

    # ------------------------------------------------------------------ measurement
    def dominant_kernel_roofline(self, stream, peak_gbs, reps=7):
        """HIP-event timing of every launch group of the decode step (eager, on the step's own stream, weights of
        all layers in turn so each launch streams from HBM).  Returns the `roofline` object of the group that
        takes the most time, plus the per-group table."""
        names = {0: "norm+qkv+rope+cache (qmm_kernel)", 1: "paged_attention_v2 (+reduce)", 2: "wo+residual (qmm_kernel)",
                 3: "norm+gate/up+silu (qmm_kernel)", 4: "down+residual (qmm_kernel)"}
        L = self.cfg.n_layers
        st = stream.cuda_stream
        ctx = getattr(self, "_ctx_now", 0)
        kv_bytes = self._batch * ctx * 2 * self.local_kv_heads * self.cfg.head_dim * 2
        rows = {}
        for part in range(5):
            evs = []
            for _ in range(reps + 1):                         # first sweep = warm-up
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
                for l in range(L):                            # back to back: the host runs ahead of the GPU
                    _check(lib.mi355_llama_run_part(self.h, l, part, st), "run_part")
                e1.record(stream)
                evs.append((e0, e1))
            torch.cuda.synchronize()
            us = sorted(a.elapsed_time(b) * 1e3 / L for a, b in evs[1:])
            avg = float(us[len(us) // 2])                     # median sweep: robust against a clock ramp in the first sweeps
            nbytes = kv_bytes if part == 1 else float(np.mean([self.part_bytes[(l, part)] for l in range(L)]))
            rows[part] = {"kernel": names[part], "avg_us": round(avg, 2), "median_us": round(us[len(us) // 2], 2),
                          "bytes": int(nbytes), "GBs": round(nbytes / avg / 1e3, 1), "launches_per_step": L}
        dom = max(rows, key=lambda p: rows[p]["avg_us"] * L)
        r = rows[dom]
        # HBM bytes per launch of the dominant kernel: NOT measurable from inside this process (PMC counters need the rocprofv3
        # wrapper), so it is quoted from the committed PMC pass of this round -- only while the mat-mul sources still hash to what that
        # pass was built from (tools/pmc_traffic.py `kernel_source_sha256`: a changed kernel makes the number stale -> null) and only
        # when that pass saw the same launch (same geometry); the pass's commit rides along in `traffic_source`
        traffic, traffic_source = None, "null: no PMC pass of these kernel sources under profiles/"
        try:
            import glob, hashlib, json, os
            root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
            csrc = os.path.join(root, "candle_vllm_amd", "csrc")
            hh = hashlib.sha256()
            for f in sorted(os.listdir(csrc)):
                if f.startswith("qmatmul") or f.startswith("qmm_") or f == "common.h":
                    hh.update(open(os.path.join(csrc, f), "rb").read())
            for path in sorted(glob.glob(os.path.join(root, "profiles", "r*_pmc_traffic.json")), reverse=True):
                pmc = json.load(open(path))
                d = pmc.get("dominant")
                if pmc.get("kernel_source_sha256") != hh.hexdigest() or not d:
                    continue
                if dom == 3 and self._batch == 1 and self.tp_world == 1 and abs(d["traffic_bytes_per_launch"] - r["bytes"]) < 0.25 * r["bytes"]:
                    traffic = int(d["traffic_bytes_per_launch"])
                    traffic_source = "profiles/%s @%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, gfx950 x2 correction; kernel sources unchanged since)" % (
                        os.path.basename(path), pmc.get("commit", "unknown"))
                break
        except Exception:
            traffic = None
        return {"bound": "hbm", "kernel": r["kernel"], "achieved": r["GBs"], "peak": peak_gbs, "unit": "GB/s",
                "frac": round(r["GBs"] / peak_gbs, 4), "traffic": traffic, "traffic_source": traffic_source, "avg_us": r["avg_us"],
                "algorithmic_bytes_per_launch": r["bytes"],
                "timing": "one hipEvent pair around the launches of all %d layers back to back on the step stream "
                          "(each streams its own weights from HBM), / %d = average launch duration of a sweep; median of %d sweeps" % (L, L, reps),
                "groups": [rows[p] for p in sorted(rows)]}

    # ------------------------------------------------------------------ KV cache (CacheEngine)
    def alloc_kv_cache(self, num_blocks):
        _check(lib.mi355_llama_alloc_kv_cache(self.h, num_blocks), "alloc_kv_cache")
        self.num_blocks = num_blocks

    def kv_shape(self):
        c = self.cfg
        if self.kv_layout == KV_FLASH:
            s = (self.num_blocks, c.block_size, self.local_kv_heads, c.head_dim)
            return s, s
        x = 16 if self.kv_layout == KV_PAGED_FP8 else 8
        return ((self.num_blocks, self.local_kv_heads, c.head_dim // x, c.block_size, x),
                (self.num_blocks, self.local_kv_heads, c.head_dim, c.block_size))

    def kv_download_u8(self, layer):
        """fp8 cache bytes (KV_PAGED_FP8)"""
        ks, vs = self.kv_shape()
        k, v = np.empty(ks, np.uint8), np.empty(vs, np.uint8)
        _check(lib.mi355_llama_kv_copy(self.h, layer, 0, k.ctypes.data, k.nbytes, 0), "kv_copy")
        _check(lib.mi355_llama_kv_copy(self.h, layer, 1, v.ctypes.data, v.nbytes, 0), "kv_copy")
        return k, v

    def kv_upload(self, layer, k_bits, v_bits):
        """k_bits/v_bits: uint16 numpy arrays (bf16 bit patterns) in the cache layout."""
        for which, a in ((0, k_bits), (1, v_bits)):
            a = np.ascontiguousarray(a, np.uint16)
            _check(lib.mi355_llama_kv_copy(self.h, layer, which, a.ctypes.data, a.nbytes, 1), "kv_copy")

    def kv_download(self, layer):
        ks, vs = self.kv_shape()
        k, v = np.empty(ks, np.uint16), np.empty(vs, np.uint16)
        _check(lib.mi355_llama_kv_copy(self.h, layer, 0, k.ctypes.data, k.nbytes, 0), "kv_copy")
        _check(lib.mi355_llama_kv_copy(self.h, layer, 1, v.ctypes.data, v.nbytes, 0), "kv_copy")
        return k, v

    def kv_fill_random(self, seed=7):
        """synthetic context: random bf16 K/V (std 1) in every block, as if a prompt had been prefetched.  Tensor parallel: every rank
        draws the GLOBAL cache (all kv heads) from the same seed and keeps its kv-head group, like `load_synthetic`."""
        from .tp import kv_head_shard
        gen = torch.Generator(device="cuda")
        gen.manual_seed(seed)
        n = lib.mi355_llama_kv_bytes_per_tensor(self.h) // 2
        Hkvg, hl = self.cfg.n_kv_heads, self.local_kv_heads
        _, kv_rank, kv_world = kv_head_shard(Hkvg, self.tp_rank, self.tp_world)
        ks, vs = self.kv_shape() if self.tp_world > 1 else (None, None)
        for l in range(self.cfg.n_layers):
            for which in (0, 1):
                if self.tp_world == 1:
                    t = torch.randn((n,), device="cuda", generator=gen).to(torch.bfloat16)
                else:
                    sh = list(ks if which == 0 else vs)
                    hax = 2 if self.kv_layout == KV_FLASH else 1              # the kv-head axis of the layout
                    sh[hax] = Hkvg
                    g = torch.randn(sh, device="cuda", generator=gen).to(torch.bfloat16)
                    t = g.narrow(hax, kv_rank * hl, hl).contiguous().view(-1)
                _check(lib.mi355_llama_kv_copy(self.h, l, which, t.data_ptr(), n * 2, 1), "kv_copy")

    # ------------------------------------------------------------------ forward
    def forward_decode(self, meta, stream=None):
        """One eager decode step.  meta: dict from oracle.ops.prepare_decode (numpy).  Returns logits (torch f32)."""
        dev = "cuda"
        B = len(meta["input_ids"])
        tok = torch.from_numpy(meta["input_ids"].astype(np.int64).astype(np.int32)).to(dev)
        pos = torch.from_numpy(meta["positions"].astype(np.int64)).to(dev)
        slots = torch.from_numpy(meta["slot_mapping"].astype(np.int64)).to(dev)
        bt = torch.from_numpy(meta["block_tables"].astype(np.int64).astype(np.int32)).contiguous().to(dev)
        ctx = torch.from_numpy(meta["context_lens"].astype(np.int64).astype(np.int32)).to(dev)
        logits = torch.empty((B, self.cfg.vocab), dtype=torch.float32, device=dev)
        st = torch.cuda.current_stream().cuda_stream if stream is None else stream
        _check(lib.mi355_llama_forward_decode(self.h, tok.data_ptr(), pos.data_ptr(), slots.data_ptr(), bt.data_ptr(),
                                              ctx.data_ptr(), B, bt.shape[1], int(meta["max_context_len"]),
                                              logits.data_ptr(), st), "forward_decode")
        torch.cuda.synchronize()
        return logits

    def forward_prefill(self, meta, stream=None):
        """One prompt step (is_prefill).  meta: dict from oracle.ops.prepare_prompt (numpy).  Returns the logits of
        every sequence's last chunk token, f32 [num_seqs, vocab]."""
        dev = "cuda"
        n = len(meta["context_lens"])
        T = len(meta["input_ids"])
        tok = torch.from_numpy(meta["input_ids"].astype(np.int64).astype(np.int32)).to(dev)
        pos = torch.from_numpy(meta["positions"].astype(np.int64)).to(dev)
        slots = torch.from_numpy(meta["slot_mapping"].astype(np.int64)).to(dev)
        bt = torch.from_numpy(meta["block_tables"].astype(np.int64).astype(np.int32)).contiguous().to(dev)
        ctx = torch.from_numpy(meta["context_lens"].astype(np.int64).astype(np.int32)).to(dev)
        cu = torch.from_numpy(meta["cu_seqlens_q"].astype(np.int64).astype(np.int32)).to(dev)
        logits = torch.empty((n, self.cfg.vocab), dtype=torch.float32, device=dev)
        st = torch.cuda.current_stream().cuda_stream if stream is None else stream
        _check(lib.mi355_llama_forward_prefill(self.h, tok.data_ptr(), pos.data_ptr(), slots.data_ptr(), bt.data_ptr(),
                                               ctx.data_ptr(), cu.data_ptr(), n, T, int(meta["max_seqlen_q"]),
                                               bt.shape[1], logits.data_ptr(), st), "forward_prefill")
        torch.cuda.synchronize()
        return logits

    def decode_begin(self, tokens, seq_lens, block_tables, ctx_cap, stream):
        tokens = np.ascontiguousarray(tokens, np.uint32)
        seq_lens = np.ascontiguousarray(seq_lens, np.uint32)
        bt = np.ascontiguousarray(block_tables, np.uint32)
        _check(lib.mi355_llama_decode_begin(self.h, tokens.ctypes.data, seq_lens.ctypes.data, bt.ctypes.data,
                                            len(tokens), bt.shape[1], int(ctx_cap), stream), "decode_begin")
        self._batch = len(tokens)
        self._ctx_now = int(np.max(seq_lens))

    def decode_step(self, stream):
        _check(lib.mi355_llama_decode_step(self.h, stream), "decode_step")
        self._ctx_now += 1

    def read_tokens(self, stream):
        out = np.empty(self._batch, np.uint32)
        _check(lib.mi355_llama_decode_read_tokens(self.h, out.ctypes.data, stream), "read_tokens")
        return out

    def graph_stats(self):
        """(step graphs captured so far, steps of the greedy loop that ran eagerly): mi355_llama_graph_captures / _eager_steps"""
        return int(lib.mi355_llama_graph_captures(self.h)), int(lib.mi355_llama_eager_steps(self.h))

    def set_attention_numerics(self, mode):
        """parity mode (tests): 1 = decode attention with the reference CPU path's bf16 rounding points (models/mod.rs:1288-1306)"""
        _check(lib.mi355_llama_set_attention_numerics(self.h, int(mode)), "set_attention_numerics")

    def set_graph(self, enable):
        """False / True, or 2 = capture tensor-parallel steps too (RCCL inside the graph; opt-in)"""
        _check(lib.mi355_llama_set_graph(self.h, 2 if enable == 2 else (1 if enable else 0)), "set_graph")

    def logits_numpy(self, batch):
        """logits of the last decode_step (device buffer of the model) -> numpy f32 [batch, vocab]"""
        torch.cuda.synchronize()
        out = np.empty((batch, self.cfg.vocab), np.float32)
        hip = ctypes.CDLL("libamdhip64.so")
        hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        hip.hipMemcpy.restype = ctypes.c_int
        _check(hip.hipMemcpy(out.ctypes.data, lib.mi355_llama_logits_ptr(self.h), out.nbytes, 2), "hipMemcpy D2H")
        return out
