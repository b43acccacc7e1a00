"""CPU: the build-time hazard check (tools/check_asm_hazards.py, run by `make hip`) finds what it is there to find -- the
two ways a hand-written asm statement bit this repo (profiles/r02_notes.md: GPU fault on address 0) -- and passes the
code objects as built."""
import glob
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("check_asm_hazards", os.path.join(ROOT, "tools", "check_asm_hazards.py"))
chk = importlib.util.module_from_spec(spec)
spec.loader.exec_module(chk)

HEAD = ["0000000000001000 <k>:"]


def test_sgpr_base_restored_by_valu_needs_five_wait_states():
    bad = HEAD + ["\tv_readlane_b32 s48, v116, 10", "\tv_readlane_b32 s49, v116, 11",
                  "\tglobal_atomic_add v81, v16, v71, s[48:49] sc0", "\ts_waitcnt vmcnt(0)"]
    f = chk.check(bad, "x.o")
    assert len(f) == 2 and "s48" in f[0] and "wait state" in f[0]
    ok = HEAD + ["\tv_readlane_b32 s48, v116, 10", "\tv_readlane_b32 s49, v116, 11", "\ts_nop 4",
                 "\tglobal_atomic_add v81, v16, v71, s[48:49] sc0", "\ts_waitcnt vmcnt(0)"]
    assert chk.check(ok, "x.o") == []
    # four other instructions in between are one wait state short, five are enough
    pad = ["\tv_add_u32_e32 v1, v2, v3"] * 4
    assert chk.check(HEAD + ["\tv_readfirstlane_b32 s8, v2"] + pad + ["\tglobal_load_dword v9, v10, s[8:9]"], "x.o")
    assert chk.check(HEAD + ["\tv_readfirstlane_b32 s8, v2"] + pad + ["\ts_nop 0", "\tglobal_load_dword v9, v10, s[8:9]"], "x.o") == []
    # an SGPR written by SALU is not this hazard
    assert chk.check(HEAD + ["\ts_mov_b32 s8, s2", "\tglobal_load_dword v9, v10, s[8:9]"], "x.o") == []


def test_returning_atomic_result_touched_before_its_wait():
    bad = HEAD + ["\tglobal_atomic_add v81, v[16:17], v71, off sc0", "\tv_mov_b32_e32 v90, v81", "\ts_waitcnt vmcnt(0)"]
    f = chk.check(bad, "x.o")
    assert len(f) == 1 and "before the s_waitcnt" in f[0]
    spill = HEAD + ["\tglobal_atomic_add v81, v[16:17], v71, off sc0", "\tscratch_store_dwordx2 off, v[80:81], s33", "\ts_waitcnt vmcnt(0)"]
    assert len(chk.check(spill, "x.o")) == 1
    ok = HEAD + ["\tglobal_atomic_add v81, v[16:17], v71, off sc0", "\tv_mov_b32_e32 v90, v82", "\ts_waitcnt vmcnt(0) lgkmcnt(0)",
                 "\tv_mov_b32_e32 v90, v81"]
    assert chk.check(ok, "x.o") == []
    # a non-returning atomic has no destination to protect
    assert chk.check(HEAD + ["\tglobal_atomic_add v16, v71, s[4:5]", "\tv_mov_b32_e32 v1, v16"], "x.o") == []


def test_built_code_objects_are_clean():
    objs = sorted(glob.glob(os.path.join(ROOT, "deltarice_amd", "csrc", "*.o")))
    if not objs:
        import pytest
        pytest.skip("run `make` first")
    total = 0
    for o in objs:
        lines = chk.disassemble(o)
        total += len(lines)
        assert chk.check(lines, os.path.basename(o)) == []
    assert total > 10000  # (the disassembly really was produced)
