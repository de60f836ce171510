#!/usr/bin/env python3
"""Lint for one code-generation fault of the compiler: a register-allocator RELOAD of a
spilled VGPR placed in a block that runs under a narrowed exec mask, directly in front of the
instruction that widens the mask again.

    .LBB_a:                        ; exit block of a divergent loop / end of an `if`: exec = the lanes
        scratch_load_dword vN ...  ;   that arrive here (none, for a loop left by s_cbranch_execz)
    .LBB_b:
        s_or_b64 exec, exec, s[..] ; the lanes that had left come back -- with vN NOT reloaded

The reload only reaches the lanes that are enabled when it executes; every other lane keeps whatever
the region used the register for.  Reads gfx950 assembly (hipcc -S --cuda-device-only), reports every
"Folded Reload" that is followed, with nothing but labels / waits / barriers / other reloads in
between, by an `s_or_b64 exec, exec, ...`, and marks the fatal ones: the block is only entered through
`s_cbranch_execz` (no lane enabled), so the reload is a no-op.  (A reload under the mask of the region
that clobbered the register -- the end of an `if` -- is correct and only listed.)

    python tools/isa_exec_lint.py k.s [kernel-name-substring]      exit code 1 if anything is found
"""
import re
import sys


def kernels(lines):
    start = None
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\S+):", l)
        if m:
            start, name = i, m.group(1)
        if start is not None and l.startswith(".Lfunc_end"):
            yield name, start, i
            start = None


NEUTRAL = re.compile(r"^\s*(s_waitcnt|s_nop|; wave barrier|s_barrier|;)")


def arrives_with_exec_zero(lines, lo, hi, i):
    """True when the block holding line i can only be entered through `s_cbranch_execz` (every lane has
    left: the exit of a divergent loop, an empty `if`) and nothing in the block before line i writes exec."""
    b = i
    while b > lo and not re.match(r"^\.LBB\d+_\d+:", lines[b]):
        if re.match(r"^\s*s_\w+ exec,|^\s*s_\w+saveexec", lines[b]):
            return False  # exec rewritten inside the block before the reload
        b -= 1
    m = re.match(r"^(\.LBB\d+_\d+):", lines[b])
    if not m:
        return False
    label = m.group(1)
    # fall-through from the block above?
    k = b - 1
    while k > lo and (not lines[k].strip() or lines[k].lstrip().startswith(";")):
        k -= 1
    # (falling out of a loop whose back edge is `s_cbranch_execnz` also means: no lane left)
    falls_in = not re.match(r"^\s*(s_branch|s_endpgm|s_setpc|s_cbranch_execnz)", lines[k])
    zero_fall = bool(re.match(r"^\s*s_cbranch_execnz", lines[k]))
    refs = [l for l in lines[lo:hi] if re.search(re.escape(label) + r"\b", l) and not l.startswith(label)]
    return (not falls_in) and (bool(refs) or zero_fall) and all(re.match(r"^\s*s_cbranch_execz", r) for r in refs)


def lint(lines, lo, hi):
    found = []
    i = lo
    while i < hi:
        l = lines[i]
        if "Folded Reload" in l and "scratch_load" in l:
            j = i + 1
            while j < hi:
                t = lines[j]
                if not t.strip() or re.match(r"^\.LBB\d+_\d+:", t) or NEUTRAL.match(t) or \
                        ("Folded Reload" in t and "scratch_load" in t):
                    j += 1
                    continue
                break
            dead = arrives_with_exec_zero(lines, lo, hi, i)
            if (j < hi and re.match(r"^\s*s_or_b64 exec, exec, s\[", lines[j])) or dead:
                found.append((i - lo, l.strip(), lines[j].strip() if j < hi else "", dead))
        i += 1
    return found


def main():
    lines = open(sys.argv[1]).read().split("\n")
    key = sys.argv[2] if len(sys.argv) > 2 else ""
    bad = sus = 0
    for name, lo, hi in kernels(lines):
        if key and key not in name:
            continue
        f = lint(lines, lo, hi)
        if f:
            sus += len(f)
            bad += sum(1 for x in f if x[3])
            print(f"{name}: {len(f)} reload(s) in front of an exec restore")
            for off, a, b, dead in f:
                print(f"    +{off}: {a}   ->   {b}" + ("   ** block entered with exec = 0: the reload reaches NO lane **" if dead else ""))
    print(f"{sus} reload(s) in front of an exec restore, {bad} of them in a block that is entered with exec = 0")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
