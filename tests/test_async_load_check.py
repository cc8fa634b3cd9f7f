"""tools/async_load_check.py on hand-written ISA: the pattern it exists for - a read of an inline-asm load's destination
registers before the next vmcnt wait, also behind an unconditional branch - is flagged; the waited-for read, LDS-DMA loads
and compiler-issued loads are not."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

ISA = """
_Z6faultyv:
	v_add_u32_e32 v2, v2, v3
	;;#ASMSTART
	global_load_dwordx4 v[130:133], v[138:139], off
	;;#ASMEND
	s_branch .LBB0_2
.LBB0_1:
	v_mov_b32_e32 v130, 0
.LBB0_2:
	v_add_f32_e32 v129, v144, v145
	v_mov_b64_e32 v[146:147], v[132:133]
	s_waitcnt vmcnt(0)
	s_endpgm
_Z5cleanv:
	;;#ASMSTART
	global_load_dwordx4 v[130:133], v[138:139], off
	;;#ASMEND
	;;#ASMSTART
	global_load_lds_dwordx4 v8, s[16:17]
	;;#ASMEND
	v_add_f32_e32 v129, v144, v145
	;;#ASMSTART
	s_waitcnt vmcnt(0)
	;;#ASMEND
	v_mov_b64_e32 v[146:147], v[132:133]
	global_load_dword v1, v[2:3], off
	v_mov_b32_e32 v5, v1
	s_endpgm
"""


def _run(tmp_path, *pattern):
    f = tmp_path / "k.s"
    f.write_text(ISA)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "async_load_check.py"), str(f), *pattern],
                       capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    return r.stdout


def test_a_copy_of_an_in_flight_load_is_flagged_and_waited_reads_are_not(tmp_path):
    out = _run(tmp_path)
    assert "2 kernels with inline-asm register loads, 2 loads, 1 read before their wait" in out
    assert "_Z6faultyv" in out and "v_mov_b64_e32 v[146:147], v[132:133]" in out
    assert "_Z5cleanv" not in out.split("\n")[0]
    assert "1 inline-asm loads, 0 read before a vmcnt wait" in _run(tmp_path, "cleanv")
    assert "1 inline-asm loads, 1 read before a vmcnt wait" in _run(tmp_path, "faultyv")
