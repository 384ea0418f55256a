#!/usr/bin/env python3
"""Build step (runs HERE, part of `make`): rewrites the packed-fp32 instructions the MI355X gets wrong next to MFMA waves.

Found r04 (profiles/r04/nondeterminism_r04.md, tools/microbench/r04_pk_opsel_sweep.hip): `v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32` whose LOW pass
selects src0's low and src1's HIGH register -- `op_sel:[0,1]` -- reads one of the two operands as 0 in lanes 48-63 while a wave of the same
SIMD has 32x32x16 MFMAs in flight (about every second instance in the reproducer; every other selection, [0,0] [1,0] [1,1], is clean in 1e9
checks).  hipcc emits that selection when it vectorises scalar code (the latent-grid tap arithmetic); it is the launch-to-launch
difference of the latent-grid kernels that r03 papered over with s_nops.  All three operations commute in src0 / src1, so the instruction
is rewritten with the two sources -- and their entries in op_sel / op_sel_hi / neg_lo / neg_hi -- exchanged: the low pass then selects [1,0].

usage: fix_pk_opsel.py in.s out.s        prints the number of rewritten instructions
       fix_pk_opsel.py --check FILE...   (.s / .dis / objects): exit code 1 if an instruction with the bad selection is present"""
import os
import re
import subprocess
import sys
import tempfile

L = "/opt/rocm/lib/llvm/bin"
INSN = re.compile(r"^(\s*)(v_pk_(?:mul|add|fma)_f32)\s+([^;/]*?)(\s*(?://|;).*)?$")
MOD = re.compile(r"(op_sel|op_sel_hi|neg_lo|neg_hi):\[([01,]+)\]")


def split_operands(text):
    """'v[0:1], v[2:3], s[4:5] op_sel:[0,1] ...' -> (['v[0:1]', 'v[2:3]', 's[4:5]'], 'op_sel:[0,1] ...')"""
    m = re.search(r"\s(op_sel|op_sel_hi|neg_lo|neg_hi|clamp)\b", text)
    ops, mods = (text[:m.start()], text[m.start():].strip()) if m else (text, "")
    return [o.strip() for o in ops.split(",")], mods


def bad_selection(mods):
    m = re.search(r"op_sel:\[([01,]+)\]", mods)
    return bool(m) and m.group(1).split(",")[:2] == ["0", "1"]


def rewrite(line):
    m = INSN.match(line)
    if not m:
        return line, False
    indent, op, rest, comment = m.group(1), m.group(2), m.group(3), m.group(4) or ""
    ops, mods = split_operands(rest)
    if not bad_selection(mods):
        return line, False
    ops[1], ops[2] = ops[2], ops[1]

    def swap(mm):
        v = mm.group(2).split(",")
        v[0], v[1] = v[1], v[0]
        return "%s:[%s]" % (mm.group(1), ",".join(v))
    return "%s%s %s %s%s" % (indent, op, ", ".join(ops), MOD.sub(swap, mods), comment), True


def device_code_objects(path, tmpdir):
    """gfx950 code objects embedded in an object file or a linked library: the .hip_fatbin section of a library holds ONE offload bundle per
    translation unit (4 KiB-aligned, each starting with the bundler's magic string); every one of them is unbundled."""
    fat = os.path.join(tmpdir, "fat.bin")
    if subprocess.run([L + "/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", path, fat], capture_output=True).returncode != 0 or not os.path.exists(fat):
        return []
    data = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(re.escape(magic), data)]
    out = []
    for i, a in enumerate(starts):
        piece = os.path.join(tmpdir, "bundle%d.bin" % i)
        open(piece, "wb").write(data[a:starts[i + 1] if i + 1 < len(starts) else len(data)])
        co = os.path.join(tmpdir, "dev%d.co" % i)
        if subprocess.run([L + "/clang-offload-bundler", "--type=o", "--input=" + piece, "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                           "--output=" + co, "--unbundle"], capture_output=True).returncode == 0 and os.path.exists(co) and os.path.getsize(co) > 0:
            out.append(co)
    return out


def disassemble(path):
    if path.endswith((".s", ".dis")):
        return open(path).read().splitlines()
    with tempfile.TemporaryDirectory() as t:
        lines = []
        for co in device_code_objects(path, t):  # (none: a host-only translation unit / library)
            lines += subprocess.run([L + "/llvm-objdump", "-d", "--no-show-raw-insn", co], capture_output=True, text=True).stdout.splitlines()
        return lines


def check_lines(lines):
    n = bad = 0
    for line in lines:
        if "v_pk_" not in line:
            continue
        m = INSN.match(line)
        if m:
            n += 1
            bad += bad_selection(split_operands(m.group(3))[1])
    return n, bad


def check(paths):
    total = 0
    for p in paths:
        n, bad = check_lines(disassemble(p))
        print("%-28s %6d packed-fp32 instructions, %d with op_sel:[0,1]" % (os.path.basename(p), n, bad))
        total += bad
    return total


def main():
    if sys.argv[1] == "--check":
        sys.exit(1 if check(sys.argv[2:]) else 0)
    n, out = 0, []
    for line in open(sys.argv[1]).read().split("\n"):
        line, changed = rewrite(line)
        n += changed
        out.append(line)
    open(sys.argv[2], "w").write("\n".join(out))
    print("fix_pk_opsel: %d instruction(s) rewritten in %s" % (n, os.path.basename(sys.argv[1])))


if __name__ == "__main__":
    main()
