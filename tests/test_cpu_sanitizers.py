"""AddressSanitizer + UndefinedBehaviorSanitizer over the HOST side of the library (no GPU): the block manager /
prefix cache / scheduler (csrc/block_engine.cpp), the GGUF reader and byte-range sharder (csrc/gguf_reader.cpp) and the
RoPE table builders (csrc/rope_tables.cpp) are plain C++, so they are rebuilt with g++ -fsanitize=address,undefined into
a stand-in library (device-side entry points become aborting stubs) and the CPU tests that drive them are re-run against
it in a child process.  Any heap error, leak-free-but-wild access, signed overflow or misaligned load in those paths
fails this test."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "candle_vllm_amd", "csrc")
HOST_SOURCES = ["block_engine.cpp", "gguf_reader.cpp", "rope_tables.cpp"]
# the CPU tests that only call host entry points; the TP loader test reaches a device entry point by design
HOST_TESTS = ["tests/test_cpu_block_engine.py", "tests/test_cpu_scheduler.py", "tests/test_cpu_gguf.py",
              "tests/test_cpu_rope_tables.py"]
DESELECT = "not load_gguf_tp and not check_gguf and not huggingface"       # loader entry points live in the device-side host_model.cpp


def _tool(name):
    r = subprocess.run(["gcc", "-print-file-name=" + name], stdout=subprocess.PIPE, text=True)
    p = r.stdout.strip()
    return p if r.returncode == 0 and os.path.isabs(p) and os.path.exists(p) else None


@pytest.mark.timeout(900)
def test_host_code_is_clean_under_asan_and_ubsan(tmp_path):
    if shutil.which("g++") is None or _tool("libasan.so") is None:
        pytest.skip("g++ / libasan not available")
    sys.path.insert(0, ROOT)
    from candle_vllm_amd import _lib
    flags = ["-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined",
             "-fno-sanitize-recover=undefined", "-shared", "-fPIC"]
    host0 = os.path.join(tmp_path, "libhost0.so")
    srcs = [os.path.join(CSRC, f) for f in HOST_SOURCES]
    r = subprocess.run(["g++"] + flags + srcs + ["-o", host0], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    nm = subprocess.run(["nm", "-D", "--defined-only", host0], stdout=subprocess.PIPE, text=True).stdout
    defined = {ln.split()[-1] for ln in nm.splitlines() if " T " in ln}
    missing = [s for s in _lib.declared_symbols() if s not in defined]
    assert "mi355_be_create" not in missing and "mi355_gguf_open" not in missing and "mi355_rope_tables" not in missing
    stubs = os.path.join(tmp_path, "stubs.c")
    with open(stubs, "w") as f:
        f.write("#include <stdlib.h>\n#include <stdio.h>\n")
        for s in missing:                                  # device-side entry points: must not be reached by these tests
            f.write('void %s(void) { fprintf(stderr, "device entry point %s reached in the host-only build\\n"); abort(); }\n' % (s, s))
    full = os.path.join(tmp_path, "libmi355vllm_asan.so")
    r = subprocess.run(["g++"] + flags + srcs + ["-x", "c", stubs, "-o", full], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    env = dict(os.environ)
    env.update({"MI355_LIB_PATH": full, "LD_PRELOAD": _tool("libasan.so"),
                # CPython's own allocations are not ours to judge; everything else is fatal
                "ASAN_OPTIONS": "detect_leaks=0:abort_on_error=0:halt_on_error=1:exitcode=97",
                "UBSAN_OPTIONS": "print_stacktrace=1:halt_on_error=1:exitcode=98"})
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", "-k", DESELECT] + HOST_TESTS,
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=800)
    tail = r.stdout[-4000:]
    assert "AddressSanitizer" not in r.stdout and "runtime error:" not in r.stdout, tail
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "failed" not in r.stdout.splitlines()[-1], tail
