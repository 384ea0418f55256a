#!/usr/bin/env python3
"""Developer check (runs HERE): hipcc's hazard recognizer does not look into inline assembly, and on gfx950 a VALU instruction of another opcode
that reads the result of a DOT instruction needs three wait states (LLVM GCNHazardRecognizer, checkMAIVALUHazards: DotWriteDifferentVALURead).
The kernels start their latent-grid sums with an inline-assembly `v_dot2_f32_f16 vD, vA, vB, 0` (srn_device.hpp dot2_from_zero); this
script disassembles every object of the build and reports any such instruction whose result is read within the next three wait states.
usage: tools/check_dot_hazard.py [object files ...]     exit code 1 if a violation is found"""
import glob
import os
import re
import subprocess
import sys
import tempfile

L = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


TRANS = re.compile(r"v_(exp|log|rcp|rcp_iflag|rsq|sqrt|sin|cos)_(f32|f16|legacy_f32)")


def regs(tok):
    m = re.fullmatch(r"-?\|?v(\d+)\|?", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"-?v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def disassemble(obj):
    """every gfx950 code object embedded in `obj` (object file or linked library), disassembled: tools/fix_pk_opsel.py"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("fix_pk_opsel", os.path.join(ROOT, "tools", "fix_pk_opsel.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m.disassemble(obj)


def check(obj, dis=None):
    bad = 0
    if dis is None:
        dis = disassemble(obj)
    kernel = "?"
    pending = []  # (register, wait states left, line)
    n_dots = 0
    for line in dis:
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            kernel = m.group(1)
            pending = []
            continue
        body = line.split("//")[0].strip()
        if not body:
            continue
        parts = body.replace(",", " ").split()
        op, args = parts[0], parts[1:]
        if op.startswith("s_nop"):
            states = int(args[0], 0) + 1
        else:
            states = 1
            if op.startswith("v_") or op.startswith("global_") or op.startswith("ds_") or op.startswith("buffer_") or op.startswith("scratch_"):
                # sources: every operand but the first (stores have no destination: all of them)
                srcs = args if ("store" in op or op.startswith("ds_write")) else args[1:]
                used = set()
                for a in srcs:
                    used |= regs(a)
                # (a write of the same register by another VALU opcode needs four wait states: one more than a read)
                wrote = regs(args[0]) if (args and not ("store" in op or op.startswith("ds_write"))) else set()
                for (r, left, src_line) in pending:
                    if r in wrote and not op.startswith("v_dot2") and left + 1 > 0 and r not in used and not TRANS.match(src_line.split()[0]):
                        print("%s: %s\n    overwrites v%d, written by `%s` with %d wait state(s) still due" % (kernel[:90], body, r, src_line, left + 1))
                        bad += 1
                for (r, left, src_line) in pending:
                    if TRANS.match(src_line.split()[0]) and (TRANS.match(op) or not op.startswith("v_")):
                        continue  # (trans -> trans and trans -> memory are not in that hazard class)
                    if r in used and not op.startswith("v_dot2_f32_f16"):
                        print("%s: %s\n    reads v%d, written by `%s` with %d wait state(s) still due" % (kernel[:90], body, r, src_line, left))
                        bad += 1
        pending = [(r, left - states, s) for (r, left, s) in pending if left - states > 0]
        # VALU trans-use hazard (gfx940 / gfx950): a non-transcendental VALU instruction that reads the result of a transcendental one
        # needs one wait state.  The compiler handles its own instructions; the inline-assembly rotations (srn_device.hpp) read feature
        # registers that v_cos / v_sin wrote, so the same scan covers them.
        if TRANS.match(op) and args:
            for r in regs(args[0]):
                pending.append((r, 1, body))
        if op == "v_dot2_f32_f16" and args and args[-1] == "0":
            n_dots += 1
            for r in regs(args[0]):
                pending.append((r, 3, body))
        if op.startswith("s_cbranch") or op.startswith("s_branch"):
            pass  # (conservative: the window continues on the fall-through path)
    return bad, n_dots


def main():
    objs = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "fv-srn_amd", "csrc", "build", "*.o")))
    total = 0
    for o in objs:
        if os.path.basename(o) in ("api.o", "pack.o", "scene_network.o"):
            continue
        bad, n = check(o)
        print("%-28s %6d inline dot2-from-zero instructions, %d read too early" % (os.path.basename(o), n, bad))
        total += bad
    sys.exit(1 if total else 0)


if __name__ == "__main__":
    main()
