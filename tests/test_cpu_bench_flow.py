"""bench.py's multi-rank CONTROL FLOW on CPU: two processes over gloo run `bench.main()` with the model handle replaced
by a stand-in whose collective-bearing methods really do a gloo collective (so a call that only one rank makes hangs
and the test times out) and with the torch.cuda entry points stubbed.  What this pins: the N > 1 launch contract
(RANK / LOCAL_RANK / WORLD_SIZE from the env, barrier + max-over-ranks timing, every rank runs the roofline sweep,
rank 0 alone prints ONE JSON line with the driver's keys, legs that only make sense at N = 1 are skipped).  The
numbers are meaningless here; the kernels, RCCL and the real handle are the GPU tests' business."""
import io
import json
import os
import socket
import sys
import types

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.multiprocessing as mp                      # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_model_module(dist, calls):
    class FakeLib:
        @staticmethod
        def mi355_set_tuning(k, v):
            return 0

    class FakeGGUFLLaMa:
        def __init__(self, cfg, max_batch=1, max_blocks_per_seq=None, kv_layout=0, tp_rank=0, tp_world=1):
            self.cfg, self.tp_rank, self.tp_world = cfg, tp_rank, tp_world
            self.weight_bytes = 4_600_000_000 // tp_world
            calls.append(("create", tp_rank, tp_world, max_batch))

        def _collective(self):                           # stands for the all-reduce / all-gather inside a TP step
            if self.tp_world > 1:
                t = torch.ones(1)
                dist.all_reduce(t)
                assert int(t.item()) == self.tp_world

        def init_comm(self, d, p2p=False, wire_bf16=False):
            t = torch.full((128,), float(self.tp_rank == 0))
            d.broadcast(t, src=0)
            assert float(t.sum()) == 128.0
            calls.append(("init_comm",))

        def comm_capture_ok(self, stream):               # every rank tests locally, bench.py then takes the MIN over ranks
            calls.append(("capture_probe",))
            return True

        def load_synthetic(self, seed=0, recipe=""):
            calls.append(("load_synthetic", recipe))

        def alloc_kv_cache(self, n):
            self.num_blocks = n

        def kv_fill_random(self, seed=0):
            pass

        def set_graph(self, on):
            calls.append(("graph", bool(on)))

        def decode_begin(self, tokens, seq_lens, bt, ctx_cap=0, stream=0):
            assert len(tokens) == len(seq_lens) == bt.shape[0]
            self._collective()

        def decode_step(self, st):
            self._collective()
            calls.append(("step",))

        def read_tokens(self, st):
            return np.zeros(1, np.uint32)

        @property
        def weight_bytes_global(self):
            return self.weight_bytes * self.tp_world

        def dominant_kernel_roofline(self, stream, peak, reps=7):
            self._collective()                           # the wo / down launch groups contain the all-reduce
            calls.append(("roofline",))
            return {"bound": "hbm", "achieved": 1.0, "peak": peak, "unit": "GB/s", "frac": 1.0 / peak, "traffic": None}

    m = types.ModuleType("candle_vllm_amd.model")
    m.GGUFLLaMa, m.lib, m.KV_PAGED, m.KV_FLASH = FakeGGUFLLaMa, FakeLib, 1, 0
    real = types.SimpleNamespace(hidden=4096, n_layers=32, n_heads=32, n_kv_heads=8, head_dim=128, intermediate=14336, vocab=128256,
                                 rms_eps=1e-5, rope_theta=500000.0, max_seq=8192, block_size=64)
    m.ModelDims = types.SimpleNamespace(llama3_8b=lambda: real)
    return m


def _worker(rank, world, port, q):
    try:
        os.environ.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world),
                           "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
        sys.path.insert(0, ROOT)
        import torch.distributed as dist
        real_init, real_tensor = dist.init_process_group, torch.tensor
        dist.init_process_group = lambda backend, **kw: real_init(
            "gloo", rank=kw["rank"], world_size=kw["world_size"])      # "nccl" + device_id on the GPU box
        torch.tensor = lambda *a, **kw: real_tensor(*a, **{k: v for k, v in kw.items() if k != "device"})
        torch.cuda.set_device = lambda d: None
        torch.cuda.synchronize = lambda *a: None
        torch.cuda.Stream = lambda *a, **kw: types.SimpleNamespace(cuda_stream=0)
        calls = []
        import candle_vllm_amd
        fake = _fake_model_module(dist, calls)
        sys.modules["candle_vllm_amd.model"] = fake
        candle_vllm_amd.model = fake
        import bench
        sys.argv = ["bench.py", "--gpus", str(world), "--steps", "3", "--warmup", "2"]
        buf, old = io.StringIO(), sys.stdout
        sys.stdout = buf
        try:
            bench.main()
        finally:
            sys.stdout = old
        q.put((rank, "ok", buf.getvalue(), calls))
    except BaseException as e:                           # SystemExit included: report, never leave the peer waiting silently
        q.put((rank, "error", repr(e), []))
        raise


@pytest.mark.timeout(300)
def test_bench_main_two_ranks_control_flow():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {}
    try:
        for _ in range(2):
            r = q.get(timeout=240)                       # a rank-0-only collective shows up here as a timeout
            assert r[1] == "ok", r
            res[r[0]] = r[2:]
    finally:
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.terminate()                            # this test's own children only
    out0, calls0 = res[0]
    out1, calls1 = res[1]
    assert out1.strip() == ""                            # only rank 0 prints
    lines = [ln for ln in out0.splitlines() if ln.strip()]
    assert len(lines) == 1
    j = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in j, key
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["warmup"] == 2 and j["scaling"] == "strong"
    assert j["vs_baseline"] is None and j["higher_is_better"] is True and j["data"] == "synthetic"
    assert j["config"]["parallelism"] == "tp2" and j["config"]["graph"] is True and "workload" in j["config"]   # TP steps are captured by default
    assert j["config"]["all_reduce"] == "RCCL on a side stream" and j["config"]["wire"] == "f32"
    assert "cpu_baseline" not in j and "batch32" not in j          # N = 1 legs
    assert j["value"] > 0 and abs(j["value"] - 1e3 / j["ms_per_step"]) / j["value"] < 1e-2      # batch 1: tokens/s = steps/s
    for calls in (calls0, calls1):
        assert ("init_comm",) in calls and ("roofline",) in calls and ("graph", True) in calls
        assert calls.count(("step",)) == 5               # 2 warm-up + 3 timed, on every rank
    assert calls0[0] == ("create", 0, 2, 1) and calls1[0] == ("create", 1, 2, 1)
