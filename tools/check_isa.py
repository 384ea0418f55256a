#!/usr/bin/env python3
"""Developer check (runs HERE, no GPU; tests/test_capi_symbols.py runs it): every object of the build is disassembled ONCE and scanned for
  * inline-assembly DOT instructions whose result is read inside the three wait states gfx950 wants (tools/check_dot_hazard.py),
  * packed-fp32 instructions with the source selection the MI355X gets wrong next to MFMA waves (tools/fix_pk_opsel.py --check).
usage: tools/check_isa.py [-j N] [object files ...]       exit code 1 if anything is found"""
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


def scan(obj):
    import contextlib
    import io
    dot, pk = _load("check_dot_hazard"), _load("fix_pk_opsel")
    lines = dot.disassemble(obj)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        bad_dot, n_dot = dot.check(obj, lines)
    n_pk, bad_pk = pk.check_lines(lines)
    return os.path.basename(obj), n_dot, bad_dot, n_pk, bad_pk, buf.getvalue()


def main():
    args = sys.argv[1:]
    jobs = 4
    if args[:1] == ["-j"]:
        jobs, args = int(args[1]), args[2:]
    objs = args or sorted(glob.glob(os.path.join(ROOT, "fv-srn_amd", "csrc", "build", "*.o")))
    total = 0
    with multiprocessing.Pool(jobs) as pool:
        for name, n_dot, bad_dot, n_pk, bad_pk, text in pool.imap(scan, objs):
            sys.stdout.write(text)
            print("%-28s %6d inline dot2-from-zero instructions, %d read too early; %6d packed-fp32 instructions, %d with op_sel:[0,1]" % (name, n_dot, bad_dot, n_pk, bad_pk))
            total += bad_dot + bad_pk
    sys.exit(1 if total else 0)


if __name__ == "__main__":
    main()
