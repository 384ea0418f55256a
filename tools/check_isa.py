#!/usr/bin/env python3
"""Developer check (runs HERE, no GPU; tests/test_capi_symbols.py runs it): every object of the build is disassembled ONCE and scanned for
  * inline-assembly DOT instructions whose result is read inside the three wait states gfx950 wants (tools/check_dot_hazard.py),
  * packed-fp32 instructions with the source selection the MI355X gets wrong next to MFMA waves (tools/fix_pk_opsel.py --check).
Without arguments the scan covers EVERYTHING that carries device code and can be loaded on a GPU box: the objects of the build, the linked
libfvsrn.so (every offload bundle in it: a translation unit that reached the link without passing fv-srn_amd/csrc/hipcc_fixed.sh shows here), the
pyrenderer extension (plain host C++: must hold no device code at all), the test helper tools/dev/bin/libaggressor.so and the microbenchmark binaries
under tools/microbench/ -- and it FAILS on any .o / .so below fv-srn_amd/ or tools/dev/bin/ that is not in that list.
usage: tools/check_isa.py [-j N] [object files ...]       exit code 1 if anything is found
       tools/check_isa.py --linked LIB OBJECT...          (the Makefile's link step) every code object in LIB comes from one of OBJECT..."""
import glob
import importlib.util
import multiprocessing
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", name + ".py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def scan(obj, dot_check=True):
    """-> (name, dot2 count, dot2 violations, packed-fp32 count, bad selections, report text, sha256 of every embedded gfx950 code object)"""
    import contextlib
    import hashlib
    import io
    import subprocess
    import tempfile
    dot, pk = _load("check_dot_hazard"), _load("fix_pk_opsel")
    lines, hashes = [], []
    with tempfile.TemporaryDirectory() as t:
        for co in pk.device_code_objects(obj, t):
            hashes.append(hashlib.sha256(open(co, "rb").read()).hexdigest())
            lines += subprocess.run([pk.L + "/llvm-objdump", "-d", "--no-show-raw-insn", co], capture_output=True, text=True).stdout.splitlines()
    buf = io.StringIO()
    bad_dot = n_dot = 0
    if dot_check:
        with contextlib.redirect_stdout(buf):
            bad_dot, n_dot = dot.check(obj, lines)
    n_pk, bad_pk = pk.check_lines(lines)
    return os.path.basename(obj), n_dot, bad_dot, n_pk, bad_pk, buf.getvalue(), hashes


def scan_pk_only(obj):
    return scan(obj, dot_check=False)


def linked_code_objects(lib):
    """sha256 of every gfx950 code object inside a linked library (no disassembly)"""
    import hashlib
    import tempfile
    pk = _load("fix_pk_opsel")
    with tempfile.TemporaryDirectory() as t:
        return [hashlib.sha256(open(co, "rb").read()).hexdigest() for co in pk.device_code_objects(lib, t)]


# Developer A/B builds (tools/variant.sh, tools/ablate.sh, `make asan`): git-ignored, never loaded by the package (FVSRN_LIBRARY points a tool at one on
# purpose); the ablation builds go through hipcc_fixed.sh as well, the sanitizer build links the device objects of the main build.
DEV_BUILD_DIRS = ("build_abl_", "build_var_", "build_asan", os.sep + "ablate" + os.sep)
# Instruction-level reproducers of the packed-fp32 erratum (profiles/r04/nondeterminism_r04.md): they must KEEP the selection they demonstrate.
REPRODUCERS = ("r04_pk_opsel_sweep", "r04_pk_raw_matrix", "r04_pk_mfma_neighbour", "r03_pk_chain")


def default_targets():
    """(objects of the build, linked libraries made of them, other binaries with device code, host-only extensions, microbenchmarks, unexpected files)"""
    pkg = os.path.join(ROOT, "fv-srn_amd")
    objs = sorted(glob.glob(os.path.join(pkg, "csrc", "build", "*.o")))
    linked = [p for p in (os.path.join(pkg, "libfvsrn.so"),) if os.path.exists(p)]
    others = [p for p in (os.path.join(ROOT, "tools", "dev", "bin", "libaggressor.so"), os.path.join(ROOT, "tools", "dev", "bin", "liboccupy.so")) if os.path.exists(p)]
    host_only = sorted(glob.glob(os.path.join(pkg, "pyrenderer", "*.so")))
    micro = [p for p in sorted(glob.glob(os.path.join(ROOT, "tools", "microbench", "*")) + glob.glob(os.path.join(ROOT, "tools", "microbench", "bin", "*")))
             if os.path.isfile(p) and os.access(p, os.X_OK) and not p.endswith((".hip", ".md", ".sh", ".py")) and os.path.basename(p) not in REPRODUCERS]
    known = set(objs + linked + others + host_only)
    unexpected = []
    for base in (pkg, os.path.join(ROOT, "tools", "dev", "bin")):
        for d, _, files in os.walk(base):
            if any(x in d + os.sep for x in DEV_BUILD_DIRS):
                continue
            for f in files:
                p = os.path.join(d, f)
                if f.endswith((".o", ".so")) and p not in known and not (f.startswith("libfvsrn_") and "asan" in f):
                    unexpected.append(p)
    return objs, linked, others, host_only, micro, unexpected


def check_linked(lib, objs):
    """Link-time check of the Makefile: every code object inside `lib` is, byte for byte, the code object of one of `objs` -- each of which
    hipcc_fixed.sh assembled from a rewritten and re-checked assembly file.  No disassembly."""
    import hashlib
    import tempfile
    pk = _load("fix_pk_opsel")
    seen = set()
    for o in objs:
        with tempfile.TemporaryDirectory() as t:
            seen.update(hashlib.sha256(open(co, "rb").read()).hexdigest() for co in pk.device_code_objects(o, t))
    hs = linked_code_objects(lib)
    foreign = [h for h in hs if h not in seen]
    print("%s: %d code objects, %d of them not from the %d objects of this build" % (os.path.basename(lib), len(hs), len(foreign), len(objs)))
    return 1 if foreign or not hs else 0


def main():
    args = sys.argv[1:]
    if args[:1] == ["--linked"]:
        sys.exit(check_linked(args[1], args[2:]))
    jobs = 4
    if args[:1] == ["-j"]:
        jobs, args = int(args[1]), args[2:]
    total = 0
    objs, linked, micro = args, [], []
    if not args:
        objs, linked, others, host_only, micro, unexpected = default_targets()
        objs = objs + others
        for p in unexpected:
            print("UNEXPECTED binary with possible device code, not part of the checked build: " + os.path.relpath(p, ROOT))
            total += 1
        pk = _load("fix_pk_opsel")
        for p in host_only:
            n = len(pk.disassemble(p))
            print("%-28s host-only extension: %s" % (os.path.basename(p)[:28], "no device code" if n == 0 else "%d lines of DEVICE CODE that did not pass the erratum pass" % n))
            total += 1 if n else 0
    seen = set()
    line = "%-28s %6d inline dot2-from-zero instructions, %d read too early; %6d packed-fp32 instructions, %d with op_sel:[0,1]"
    with multiprocessing.Pool(jobs) as pool:
        for name, n_dot, bad_dot, n_pk, bad_pk, text, hashes in pool.imap(scan, objs):
            sys.stdout.write(text)
            print(line % (name, n_dot, bad_dot, n_pk, bad_pk))
            total += bad_dot + bad_pk
            seen.update(hashes)
        # microbenchmarks are timing skeletons (their results feed nothing): only the erratum's selection is looked for
        for name, _, _, n_pk, bad_pk, _, _ in pool.imap(scan_pk_only, micro):
            print("%-28s microbenchmark: %6d packed-fp32 instructions, %d with op_sel:[0,1]" % (name, n_pk, bad_pk))
            total += bad_pk
    # a linked library is clean iff every code object in it is one of the objects scanned above (same bytes): no second disassembly of 1e6 lines
    for lib in linked:
        hs = linked_code_objects(lib)
        foreign = [h for h in hs if h not in seen]
        print("%-28s %d code objects, %d of them NOT from the scanned objects" % (os.path.basename(lib), len(hs), len(foreign)))
        if foreign or not hs:
            name, n_dot, bad_dot, n_pk, bad_pk, text, _ = scan(lib)
            sys.stdout.write(text)
            print(line % (name, n_dot, bad_dot, n_pk, bad_pk))
            total += bad_dot + bad_pk + (0 if hs else 1)
    sys.exit(1 if total else 0)


if __name__ == "__main__":
    main()
