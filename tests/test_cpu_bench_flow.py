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


def _worker(rank, world, port, q):
    try:
        os.environ.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world),
                           "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
        sys.path.insert(0, ROOT)
        calls = []
        from tests import bench_standin
        bench_standin.install(calls)
        import bench
        sys.argv = ["bench.py", "--gpus", str(world), "--steps", "3", "--warmup", "2", "--settle-steps", "4"]
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
    assert j["config"]["all_reduce"] == "stand-in transport" and j["config"]["ranks"] == 2 and j["config"]["wire"] == "f32"
    assert "cpu_baseline" not in j and "batch32" not in j          # N = 1 legs
    assert j["value"] > 0 and abs(j["value"] - 1e3 / j["ms_per_step"]) / j["value"] < 1e-2      # batch 1: tokens/s = steps/s
    for calls in (calls0, calls1):
        assert ("init_comm", "auto") in calls and ("roofline",) in calls and ("graph", True) in calls
        assert calls.count(("step",)) == 4 + 3 * 5 + 5   # 4 settle steps, 3 blocks of (2 warm-up + 3 timed), 5 untimed steps for `first_tokens`, on every rank
    assert calls0[0] == ("create", 0, 2, 1) and calls1[0] == ("create", 1, 2, 1)


@pytest.mark.timeout(300)
def test_bench_py_gpus_2_spawns_its_own_ranks():
    """`python bench.py --gpus 2` as a PLAIN subprocess (what the driver runs when it does not wrap the call in torch.distributed.run
    itself; VERDICT r4: it used to exit with "launch with ..."): the entry script re-executes itself under torch.distributed.run on
    127.0.0.1, one process per rank; rank 0 prints the one JSON line with n_gpus, the transport of the all-reduce and the rank count
    it saw.  Here the entry script is the stand-in (gloo, no GPU): the spawning, the env contract and the line are bench.py's own."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "bench_standin.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--settle-steps", "2"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=280, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["warmup"] == 1 and j["scaling"] == "strong"
    assert j["config"]["parallelism"] == "tp2" and j["config"]["ranks"] == 2 and j["config"]["all_reduce"] == "stand-in transport"
    # a world size that contradicts --gpus is refused, not silently benchmarked
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"),
                        stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120, cwd=ROOT)
    assert r2.returncode != 0 and "WORLD_SIZE=1" in (r2.stderr + r2.stdout)
