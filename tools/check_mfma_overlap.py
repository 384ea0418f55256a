#!/usr/bin/env python3
"""Developer check (runs HERE, no GPU): MFMA instructions whose destination registers overlap their A or B source registers.

Found r04 (profiles/r04/nondeterminism_r04.md): hipcc 7.2 allocates `v_mfma_f32_32x32x16_f16 v[2:17], v[10:13], v[2:5], 0` -- destination on top
of both multiplicands -- for MFMAs whose C operand is the constant 0.  With two waves per SIMD such an instruction now and then reads part of
a multiplicand (the quarter of the wave that is read last: lanes 48-63) AFTER its own first result rows were written there.  The device code
therefore gives every zero-C MFMA a register C operand or keeps the multiplicands live behind it (srn_device.hpp, mfma_zero_c); this script
disassembles the build and fails on any MFMA with such an overlap.
usage: tools/check_mfma_overlap.py [object files | .s | .dis ...]     exit code 1 if an overlap is found"""
import glob
import os
import re
import subprocess
import sys
import tempfile

L = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def regs(tok):
    m = re.fullmatch(r"([av])\[(\d+):(\d+)\]", tok)
    if m:
        return m.group(1), set(range(int(m.group(2)), int(m.group(3)) + 1))
    m = re.fullmatch(r"([av])(\d+)", tok)
    if m:
        return m.group(1), {int(m.group(2))}
    return None


def disassemble(path):
    if path.endswith((".s", ".dis")):
        return open(path).read().splitlines()
    with tempfile.TemporaryDirectory() as t:
        subprocess.check_call([L + "/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", path, t + "/fat.bin"])
        subprocess.check_call([L + "/clang-offload-bundler", "--type=o", "--input=" + t + "/fat.bin", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                               "--output=" + t + "/dev.co", "--unbundle"])
        return subprocess.run([L + "/llvm-objdump", "-d", "--no-show-raw-insn", t + "/dev.co"], capture_output=True, text=True).stdout.splitlines()


def check(path, verbose):
    kernel, n, bad, kernels = "?", 0, 0, {}
    for line in disassemble(path):
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line) or re.match(r"^(_Z\w+):", line)
        if m:
            kernel = m.group(1)
            continue
        body = line.split("//")[0].split(";")[0].strip()
        if not body.startswith(("v_mfma", "v_smfmac")):
            continue
        ops = [regs(x) for x in body.replace(",", " ").split()[1:5]]
        n += 1
        d = ops[0]
        for what, x in (("A", ops[1]), ("B", ops[2])):
            if d and x and d[0] == x[0] and d[1] & x[1]:
                bad += 1
                kernels[kernel] = kernels.get(kernel, 0) + 1
                if verbose:
                    print("%s: %s   (destination overlaps %s)" % (kernel[:80], body, what))
                break
    return n, bad, kernels


def main():
    args = [a for a in sys.argv[1:] if a != "-v"]
    verbose = "-v" in sys.argv[1:]
    objs = args or sorted(glob.glob(os.path.join(ROOT, "fv-srn_amd", "csrc", "build", "*.o")))
    total = 0
    for o in objs:
        if os.path.basename(o) in ("api.o", "pack.o", "scene_network.o"):
            continue
        n, bad, kernels = check(o, verbose)
        print("%-28s %6d MFMA instructions, %4d with the destination on a multiplicand (%d kernels)" % (os.path.basename(o), n, bad, len(kernels)))
        total += bad
    sys.exit(1 if total else 0)


if __name__ == "__main__":
    main()
