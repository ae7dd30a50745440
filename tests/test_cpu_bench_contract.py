"""Static checks of bench.py's multi-rank contract (no GPU): anything that issues tensor-parallel collectives must
run on EVERY rank -- a call under `if rank == 0:` leaves rank 0 waiting for its peers and the N>1 run never prints."""
import ast
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# methods of the model handle whose launches contain the all-reduce / all-gather when tp_world > 1
COLLECTIVE_CALLS = {"dominant_kernel_roofline", "decode_step", "read_tokens", "decode_begin", "init_comm",
                    "forward_prefill", "barrier", "all_reduce"}


def _is_rank0_test(node):
    t = node.test
    return (isinstance(t, ast.Compare) and isinstance(t.left, ast.Name) and t.left.id == "rank"
            and len(t.ops) == 1 and isinstance(t.ops[0], ast.Eq)
            and isinstance(t.comparators[0], ast.Constant) and t.comparators[0].value == 0)


def _calls(node):
    for n in ast.walk(node):
        if isinstance(n, ast.Call) and isinstance(n.func, ast.Attribute):
            yield n.func.attr, n


def _guarded_by_world1(call, if_node):
    """true if, inside the rank-0 block, the call sits under a condition that mentions world == 1 / do_b32"""
    for n in ast.walk(if_node):
        if isinstance(n, ast.If) and n is not if_node and any(c is call for c in ast.walk(n)):
            src = ast.unparse(n.test)
            if "world == 1" in src or "do_b32" in src:
                return True
    return False


def test_no_collective_under_rank0_only():
    tree = ast.parse(open(os.path.join(ROOT, "bench.py")).read())
    main = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "main")
    bad = []
    for node in ast.walk(main):
        if isinstance(node, ast.If) and _is_rank0_test(node):
            for name, call in _calls(ast.Module(body=node.body, type_ignores=[])):
                if name in COLLECTIVE_CALLS and not _guarded_by_world1(call, node):
                    bad.append((name, call.lineno))
    assert not bad, f"collective-issuing calls that only rank 0 would make: {bad}"


def test_bench_line_keys_present_in_source():
    src = open(os.path.join(ROOT, "bench.py")).read()
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "workload", "roofline", "cpu_baseline"):
        assert f'"{key}"' in src, key
