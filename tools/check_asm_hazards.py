#!/usr/bin/env python3
"""Build-time check of the gfx950 code objects for two hazards that live inside hand-written asm statements, where the
compiler neither pads wait states nor tracks outstanding memory operations (round 2: a GPU fault on address 0,
profiles/r02_notes.md):

  1. an SGPR written by a VALU instruction (v_readlane / v_readfirstlane / a VOP3 compare) needs FIVE wait states before a
     vector-memory instruction reads it as (part of) its base or offset.  Checked for every vector-memory instruction of the
     disassembly: compiler-generated code always satisfies it, so any hit is an asm statement.
  2. the destination VGPR of a RETURNING global atomic (`... sc0`) must not be read or written before the next
     `s_waitcnt vmcnt(..)`: the compiler believes the asm's output is valid at once.

usage: check_asm_hazards.py file.o [file.o ...]      (objects built by hipcc; exits 1 on a finding)
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = os.environ.get("ROCM_LLVM", "/opt/rocm/lib/llvm/bin")
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"


def disassemble(obj):
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, "fat.bin"), os.path.join(td, "dev.co")
        subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fat], check=True)
        if not os.path.exists(fat) or os.path.getsize(fat) == 0:
            return []
        subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--targets=" + TARGET, "--input=" + fat,
                        "--output=" + co, "--unbundle"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        out = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co], check=True,
                             capture_output=True, text=True).stdout
    return out.splitlines()


def sgprs(tok):
    """SGPR numbers named by one operand token: s5, s[4:5]; vcc / exec are not spill targets and are left out."""
    m = re.fullmatch(r"s(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"s\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def vgprs(tok):
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


VMEM = re.compile(r"^(global_|buffer_|flat_|scratch_)")


def check(lines, name):
    findings = []
    func = "?"
    # (sgpr -> wait states since a VALU wrote it); pruned beyond 5
    recent = {}
    pending = None  # (dest vgprs, line text) of a returning atomic not yet waited for
    for ln in lines:
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", ln.strip())
        if m:
            func, recent, pending = m.group(1), {}, None
            continue
        txt = ln.split("//")[0].strip()
        if not txt or txt.endswith(":"):
            continue
        parts = txt.replace(",", " ").split()
        op, args = parts[0], parts[1:]
        # --- rule 2
        if pending is not None:
            # only a wait for EVERYTHING outstanding proves the atomic has returned: `vmcnt(N)` with N > 0 may be a wait for
            # loads issued before it.  (Labels and branches do not clear `pending`: the scan stays conservative across
            # basic blocks, which is where the kernels' uses of the ticket sit.)
            if op == "s_waitcnt" and re.search(r"vmcnt\(0\)", txt):
                pending = None
            elif op in ("s_endpgm",):
                pending = None
            else:
                used = set()
                for a in args:
                    used |= vgprs(a)
                if used & pending[0]:
                    findings.append(f"{name}: {func}: `{txt}` touches v{sorted(pending[0])} before the s_waitcnt vmcnt of `{pending[1]}`")
                    pending = None
        if op.startswith("global_atomic") and " sc0" in (" " + txt) and args and vgprs(args[0]) and len(args) >= 4:
            pending = (vgprs(args[0]), txt)
        # --- rule 1
        if VMEM.match(op):
            for a in args:
                for s in sgprs(a):
                    if s in recent and recent[s] < 5:
                        findings.append(f"{name}: {func}: `{txt}` reads s{s} {recent[s]} wait state(s) after a VALU wrote it (5 needed)")
        states = 1
        if op == "s_nop":
            states = int(args[0], 0) + 1
        recent = {s: w + states for s, w in recent.items() if w + states < 6}
        if op.startswith("v_") and args:
            dst = sgprs(args[0])
            if op.startswith("v_cmp") and len(args) >= 3:
                dst = sgprs(args[0])
            for s in dst:
                recent[s] = 0
    return findings


def main():
    objs = sys.argv[1:]
    if not objs:
        print(__doc__)
        return 2
    bad = []
    n_lines = 0
    for o in objs:
        lines = disassemble(o)
        n_lines += len(lines)
        bad += check(lines, os.path.basename(o))
    for b in bad:
        print("asm hazard:", b)
    print(f"check_asm_hazards: {len(objs)} objects, {n_lines} lines of gfx950 disassembly, {len(bad)} finding(s)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
