#!/usr/bin/env python3
"""Developer tool (hazard bisect, r04): inserts `s_nop`s into a saved device assembly file (hipcc -save-temps) so that the compiler's
schedule stays frozen while single instruction distances change.

usage: patch_s.py in.s out.s [--func REGEX] [--lines A:B] RULE [RULE ...]
RULE = before@@REGEX@@TEXT | after@@REGEX@@TEXT | at@@LINENO@@TEXT      (TEXT: `;` separates instructions, e.g. "s_nop 7;s_nop 7")
  --func   only inside functions whose (mangled) name matches
  --lines  only instructions whose 1-based line number in in.s lies in [A, B]
Prints the number of insertions per rule."""
import re
import sys


def main():
    args = sys.argv[1:]
    src, dst = args[0], args[1]
    func_re, lo, hi, rules = None, 0, 1 << 60, []
    i = 2
    while i < len(args):
        a = args[i]
        if a == "--func":
            func_re = re.compile(args[i + 1]); i += 2; continue
        if a == "--lines":
            lo, hi = (int(v) for v in args[i + 1].split(":")); i += 2; continue
        kind, pat, text = a.split("@@", 2)
        rules.append([kind, (int(pat) if kind == "at" else re.compile(pat)), ["\t" + t.strip() for t in text.split(";")], 0])
        i += 1
    out, func, active = [], None, func_re is None
    for n, line in enumerate(open(src).read().split("\n"), 1):
        m = re.match(r"^([A-Za-z_][\w$.]*):", line)
        if m and not line.startswith(".L"):
            func = m.group(1)
            active = func_re is None or bool(func_re.search(func))
        s = line.strip()
        is_insn = bool(s) and not s.startswith((";", ".", "//")) and not s.endswith(":")
        pre, post = [], []
        if active and is_insn and lo <= n <= hi:
            for r in rules:
                if (r[0] == "at" and r[1] == n) or (r[0] != "at" and r[1].search(s)):
                    (post if r[0] == "after" else pre).extend(r[2]); r[3] += 1
        elif active and not is_insn:
            for r in rules:
                if r[0] == "at" and r[1] == n:
                    pre.extend(r[2]); r[3] += 1
        out.extend(pre); out.append(line); out.extend(post)
    open(dst, "w").write("\n".join(out))
    for r in rules:
        print("  %s:%s -> %d insertions" % (r[0], r[1] if r[0] == "at" else r[1].pattern, r[3]))


if __name__ == "__main__":
    main()
