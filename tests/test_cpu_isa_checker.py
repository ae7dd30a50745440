"""tools/check_isa_vmcnt.py on hand-written gfx950 assembly: the queue model (loads retire in order, `s_waitcnt vmcnt(N)` retires all but
the N issued last) must flag a register touched while a load into it is in flight -- the failure mode of asm loads with VGPR destinations
that round 4 ran into -- and stay silent on correctly counted code, across a loop's back edge too."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "check_isa_vmcnt.py")

GOOD = """
_Z4goodv:
	global_load_dwordx4 v[0:3], v[20:21], off
	global_load_dwordx4 v[4:7], v[20:21], off offset:16
.LBB0_1:
	global_load_dwordx4 v[8:11], v[20:21], off offset:32
	s_waitcnt vmcnt(2)
	v_add_u32_e32 v30, v0, v1
	global_load_dwordx4 v[0:3], v[20:21], off offset:48
	s_waitcnt vmcnt(2)
	v_add_u32_e32 v30, v4, v30
	global_load_dwordx4 v[4:7], v[20:21], off offset:64
	s_waitcnt vmcnt(2)
	v_add_u32_e32 v30, v8, v30
	s_cbranch_scc1 .LBB0_1
	s_waitcnt vmcnt(0)
	s_endpgm
"""

# the copy at the loop's end reads v[8:11] while the load into it (issued in this trip) is still in flight
BAD = """
_Z3badv:
	global_load_dwordx4 v[0:3], v[20:21], off
.LBB0_1:
	global_load_dwordx4 v[8:11], v[20:21], off offset:32
	s_waitcnt vmcnt(1)
	v_add_u32_e32 v30, v0, v1
	v_mov_b32_e32 v0, v8
	s_cbranch_scc1 .LBB0_1
	s_waitcnt vmcnt(0)
	s_endpgm
"""

# an LDS DMA has no destination register: only its address registers count while it is being issued
DMA = """
_Z3dmav:
.LBB0_1:
	global_load_lds_dwordx4 v[2:3], off
	v_add_u32_e32 v2, 16, v2
	s_waitcnt vmcnt(1)
	s_barrier
	s_cbranch_scc1 .LBB0_1
	s_endpgm
"""


def _run(tmp_path, text, sym):
    f = tmp_path / (sym + ".s")
    f.write_text(text)
    r = subprocess.run([sys.executable, TOOL, str(f), sym], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    return r.returncode, r.stdout


def test_counted_prefetch_passes(tmp_path):
    rc, out = _run(tmp_path, GOOD, "good")
    assert rc == 0 and " 0 hazards" in out, out


def test_copy_of_a_register_in_flight_is_flagged(tmp_path):
    rc, out = _run(tmp_path, BAD, "bad")
    assert rc == 1 and "v_mov_b32_e32 v0, v8" in out, out


def test_lds_dma_has_no_register_destination(tmp_path):
    rc, out = _run(tmp_path, DMA, "dma")
    assert rc == 0, out


def test_no_vector_alu_instruction_is_hidden_in_an_asm_string():
    """hipcc's hazard pass pads wait states between a matrix instruction and the VALU instructions around it, but not through an inline-asm
    string.  A `v_cvt_pk_bf16_f32` written that way cost two defects (wrong sums behind it, round 3; the matrix core's late write landing on
    its result, the round 5-6 NaN -- common.h: cvt_pk_bf16), so the kernels keep every VALU instruction visible to the compiler: asm strings
    hold only waits / barriers / LDS and DMA traffic whose completion the code counts by hand (tools/check_isa_vmcnt.py checks those)."""
    import glob
    import re
    csrc = os.path.join(ROOT, "candle_vllm_amd", "csrc")
    hits = []
    for path in sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.inc")) + glob.glob(os.path.join(csrc, "*.h"))):
        for no, line in enumerate(open(path), 1):
            code = line.split("//")[0]
            if "asm" not in code:
                continue
            for text in re.findall(r'"([^"]*)"', code):
                if re.search(r"\bv_[a-z0-9_]+\b", text):
                    hits.append("%s:%d: %s" % (os.path.basename(path), no, text))
    assert not hits, hits
