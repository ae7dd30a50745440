import os
import sys

import pytest

# idle OpenMP threads that spin burn the CPU quota of a container whose quota is below its visible CPU count (the C
# oracle's parallel loops then run 10-50x slower); must be in the environment before libgomp is loaded
os.environ.setdefault("OMP_WAIT_POLICY", "passive")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def lib():
    """The HIP C-ABI library; built on demand (hipcc cross-compiles without a GPU).  MI355_TUNING="key:value,..." sets A/B switches for
    the whole session before the first test (how a candidate default is run through the entire suite before it becomes one, e.g.
    MI355_TUNING=44:3,5:64 = the balanced LDS-DMA attention stream wherever its shapes fit); tests that scope a key with
    candle_vllm_amd.tuning() still get their own value inside the block and this one back afterwards."""
    import __graft_entry__ as ge
    ge.build()
    import candle_vllm_amd
    for kv in filter(None, os.environ.get("MI355_TUNING", "").split(",")):
        k, v = kv.split(":")
        candle_vllm_amd.lib.mi355_set_tuning(int(k), int(v))
    return candle_vllm_amd.lib


# tests of code that exists in probe builds only (tools/build_probe_lib.sh: -DMI355_QMM_PROBES) are collected only when such a
# library is loaded:  MI355_LIB_PATH=build_probe/libmi355vllm_probes.so MI355_PROBE_BUILD=1 python -m pytest tests/probe -m gpu
import os as _os
collect_ignore_glob = [] if _os.environ.get("MI355_PROBE_BUILD") else ["probe/*"]
