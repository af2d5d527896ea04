#!/usr/bin/env python
"""
Resolve compile-time switches in the kernel sources as if they were never defined (a small `unifdef -U`): the branches of
`#ifdef X / #ifndef X / #if defined(X) / #elif defined(X) / #else / #endif` that a build WITHOUT -DX compiles stay, the others and the
directives go. Used once in round 6 to move the experiment switches whose same-box A/B was lost (docs/HISTORY.md, "Round 5: experiment
log") out of csrc/paths_team.hpp and csrc/ltpl_hip.hip; the removed text is kept as a patch under tools/experiments/ (apply with
`git apply` to get a switch back). The generated code of the default build must not change: compare __graft_entry__.build_stamp before / after.

    python tools/strip_switches.py <file> SWITCH [SWITCH ...]        (rewrites <file> in place, prints the number of directives resolved)
"""
import re
import sys


def strip(text, names):
    names = set(names)
    out = []
    # stack entries: [kind, emitting_before, taken, active] with kind 'ours' (a switch being resolved) or 'other' (left untouched)
    stack = []
    n_resolved = 0

    def emitting():
        return all(e[3] for e in stack if e[0] == 'ours')

    for line in text.split("\n"):
        m = re.match(r"\s*#\s*(ifdef|ifndef|if|elif|else|endif)\b(.*)", line)
        if not m:
            if emitting():
                out.append(line)
            continue
        d, rest = m.group(1), m.group(2).split("//")[0].strip()
        if d in ("ifdef", "ifndef", "if"):
            name = None
            if d in ("ifdef", "ifndef"):
                name = rest.split()[0] if rest.split() else None
            else:
                mm = re.fullmatch(r"(!?)\s*defined\s*\(\s*(\w+)\s*\)", rest)
                if mm:
                    name, d = mm.group(2), ("ifndef" if mm.group(1) else "ifdef")
            if name in names:
                active = d == "ifndef"            # the switch is undefined
                stack.append(['ours', emitting(), active, active])
                n_resolved += 1
            else:
                stack.append(['other', emitting(), True, True])
                if emitting():
                    out.append(line)
            continue
        if not stack:
            raise ValueError("unbalanced directive: " + line)
        top = stack[-1]
        if top[0] == 'other':
            if emitting():
                out.append(line)
            if d == "endif":
                stack.pop()
            continue
        if d == "elif":
            mm = re.fullmatch(r"(!?)\s*defined\s*\(\s*(\w+)\s*\)", rest)
            if not mm or mm.group(2) not in names:
                raise ValueError("cannot resolve '#elif %s' behind a resolved switch" % rest)
            cond = bool(mm.group(1))               # !defined(X) with X undefined -> true
            top[3] = (not top[2]) and cond
            top[2] = top[2] or top[3]
        elif d == "else":
            top[3] = not top[2]
            top[2] = True
        else:
            stack.pop()
    if stack:
        raise ValueError("unterminated conditional")
    return "\n".join(out), n_resolved


if __name__ == "__main__":
    path, names = sys.argv[1], sys.argv[2:]
    src = open(path).read()
    new, n = strip(src, names)
    open(path, "w").write(new)
    print("%s: %d conditionals resolved, %d -> %d lines" % (path, n, src.count("\n") + 1, new.count("\n") + 1))
