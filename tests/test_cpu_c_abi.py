"""The drop-in boundary as a C consumer sees it (no GPU): include/mi355_vllm.h must be valid C11 on its own -- cgo,
bindgen and a C host parse it as C, not C++ -- and a plain-C program linked against libmi355vllm.so must see the same struct
layouts the library was built with and get sane answers from host-only entry points."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INC = os.path.join(ROOT, "include")
SRC = os.path.join(ROOT, "tests", "c_abi", "consumer.c")


def _need_gcc():
    if shutil.which("gcc") is None:
        pytest.skip("gcc not on PATH")


def test_header_is_valid_c11_and_cxx17(tmp_path):
    _need_gcc()
    tu = os.path.join(tmp_path, "hdr.c")
    open(tu, "w").write('#include "mi355_vllm.h"\nint main(void) { return 0; }\n')
    for cmd in (["gcc", "-std=c11", "-pedantic"], ["g++", "-std=c++17", "-x", "c++"]):
        r = subprocess.run(cmd + ["-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I", INC, tu],
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode == 0, r.stdout


def test_plain_c_program_links_and_agrees_with_the_library(lib, tmp_path):
    _need_gcc()
    import __graft_entry__ as ge
    libdir = os.path.dirname(ge.LIB)
    exe = os.path.join(tmp_path, "consumer")
    r = subprocess.run(["gcc", "-std=c11", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", INC, SRC, "-o", exe,
                        "-L", libdir, "-lmi355vllm", "-Wl,--allow-shlib-undefined", "-Wl,-rpath," + libdir],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = "/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env, timeout=120)
    assert r.returncode == 0, r.stdout
    kv = {ln.split()[0]: ln.split()[1:] for ln in r.stdout.splitlines() if ln.strip()}
    for name in ("sizeof_qmm_desc", "sizeof_llama_config", "sizeof_dense_config", "sizeof_rope_scaling"):
        assert kv[name][0] == kv[name][1], (name, kv[name])             # C's sizeof == the library's
    from candle_vllm_amd import _lib
    import ctypes
    assert int(kv["sizeof_qmm_desc"][0]) == ctypes.sizeof(_lib.QmmDesc)          # ... == the ctypes mirror
    assert int(kv["sizeof_llama_config"][0]) == ctypes.sizeof(_lib.LlamaConfig)
    assert int(kv["sizeof_dense_config"][0]) == ctypes.sizeof(_lib.DenseConfig)
    assert int(kv["sizeof_rope_scaling"][0]) == ctypes.sizeof(_lib.RopeScaling)
    assert kv["unknown_struct"] == ["-1"]
    assert kv["repacked_q4k"] == [str(2 * 2 * 2304)] and kv["repacked_q6k"] == [str(2 * 2 * 3360)]
    assert int(kv["repacked_bad"][0]) < 0
    assert kv["rope_len"] == ["4"] and kv["rope_rc"] == ["0"] and kv["rope_row0"] == ["1.0", "0.0"]
    assert kv["gguf_open_missing"] == ["1"]


def test_every_entry_point_cites_the_reference_interface_it_replaces():
    """the header is the drop-in boundary: every declared function carries (itself, or the section banner it sits under) a
    reference citation of the form file.rs:line / file.py:line"""
    import re
    src = open(os.path.join(INC, "mi355_vllm.h")).read()
    decl = re.compile(r"^\s*(?:(?:const )?void\*?|int|int32_t|int64_t|uint64_t|float\*)\s+(\w+)\s*\(", re.M)
    cite = re.compile(r"\.(rs|py):\d+")
    last_end, section_cited, missing, n = 0, False, [], 0
    for m in decl.finditer(src):
        n += 1
        between = src[last_end:m.start()]
        end = src.index(";", m.start()) + 1
        if re.search(r"/\* -{20,}", between):
            section_cited = bool(cite.search(between[between.rfind("/* ---"):]))
        if not cite.search(between + src[m.start():end]) and not section_cited:
            missing.append(m.group(1))
        last_end = end
    assert n > 150 and not missing, missing
