"""Test infrastructure (tests/test_cpu_bench_flow.py): bench.py's N > 1 control flow on CPU.  `install()` replaces the model handle by a
stand-in whose collective-bearing methods really do a gloo collective (a call that only one rank makes hangs, and the test times out),
stubs the torch.cuda entry points and maps the "nccl" process group to gloo.  Run as a script it is bench.py's entry point with the
stand-in installed -- `python tests/bench_standin.py --gpus 2` exercises bench.py's own rank spawning (it re-executes sys.argv[0])."""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fake_model_module(dist, calls):
    class FakeLib:
        @staticmethod
        def mi355_set_tuning(k, v):
            return 0

        @staticmethod
        def mi355_comm_unique_id(ptr):                    # bench.py's pre-flight: "librccl loads"
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

        def init_comm(self, d, p2p="auto", wire_bf16=False):
            t = torch.full((128,), float(self.tp_rank == 0))
            d.broadcast(t, src=0)
            assert float(t.sum()) == 128.0
            calls.append(("init_comm", p2p))
            return "stand-in transport"

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

        def graph_stats(self):
            return (1, 1)

        @property
        def weight_bytes_global(self):
            return self.weight_bytes * self.tp_world

        def dominant_kernel_roofline(self, stream, peak, reps=7):
            self._collective()                           # the wo / down launch groups contain the all-reduce
            calls.append(("roofline",))
            return {"bound": "hbm", "achieved": 1.0, "peak": peak, "unit": "GB/s", "frac": 1.0 / peak, "traffic": None}

    m = types.ModuleType("candle_vllm_amd.model")
    m.GGUFLLaMa, m.lib, m.KV_PAGED, m.KV_FLASH = FakeGGUFLLaMa, FakeLib, 1, 0

    def comm_all_min(d, v, group=None):                  # the real helper's arithmetic on host tensors
        f = torch.tensor([int(v)], dtype=torch.int32)
        d.all_reduce(f, op=d.ReduceOp.MIN, group=group)
        return int(f.item())
    m.comm_all_min, m._ctl_device = comm_all_min, (lambda d: "cpu")
    real = types.SimpleNamespace(hidden=4096, n_layers=32, n_heads=32, n_kv_heads=8, head_dim=128, intermediate=14336, vocab=128256,
                                 rms_eps=1e-5, rope_theta=500000.0, max_seq=8192, block_size=64)
    m.ModelDims = types.SimpleNamespace(llama3_8b=lambda: real)
    return m



def install(calls):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    real_init, real_tensor = dist.init_process_group, torch.tensor
    dist.init_process_group = lambda backend, **kw: real_init(
        "gloo", rank=kw["rank"], world_size=kw["world_size"])      # "nccl" + device_id on the GPU box
    torch.tensor = lambda *a, **kw: real_tensor(*a, **{k: v for k, v in kw.items() if k != "device"})
    torch.cuda.set_device = lambda d: None
    torch.cuda.synchronize = lambda *a: None
    torch.cuda.Stream = lambda *a, **kw: types.SimpleNamespace(cuda_stream=0)
    import candle_vllm_amd
    fake = _fake_model_module(dist, calls)
    sys.modules["candle_vllm_amd.model"] = fake
    candle_vllm_amd.model = fake


if __name__ == "__main__":
    _calls = []
    install(_calls)
    import bench
    bench.main()
