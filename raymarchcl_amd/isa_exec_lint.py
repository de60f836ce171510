#!/usr/bin/env python3
"""Lint for one code-generation fault of the compiler (DESIGN_HISTORY.md 4c): a register-allocator RELOAD of a
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


LABEL = re.compile(r"^(\.LBB\d+_\d+):")
EXEC_WRITE = re.compile(r"^\s*s_\w+ exec,|^\s*s_\w+saveexec")
_cache = {}


def zero_entry_blocks(lines, lo, hi):
    """Line numbers of the labels of all blocks that can only be entered with exec = 0: every branch to
    the label is an `s_cbranch_execz`, and the block above does not fall in with lanes enabled (it ends in
    an unconditional branch, or in the `s_cbranch_execnz` back edge of a loop -- falling out of that also
    means no lane is left)."""
    key = (id(lines), lo, hi)
    if key in _cache:
        return _cache[key]
    refs = {}
    for l in lines[lo:hi]:
        m = re.match(r"^\s*(s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)", l)
        if m:
            refs.setdefault(m.group(2), []).append(m.group(1))
    out = set()
    for b in range(lo, hi):
        m = LABEL.match(lines[b])
        if not m:
            continue
        k = b - 1
        while k > lo and (not lines[k].strip() or lines[k].lstrip().startswith(";")):
            k -= 1
        falls_in = not re.match(r"^\s*(s_branch|s_endpgm|s_setpc|s_cbranch_execnz)", lines[k])
        zero_fall = bool(re.match(r"^\s*s_cbranch_execnz", lines[k]))
        r = refs.get(m.group(1), [])
        if (not falls_in) and (r or zero_fall) and all(x == "s_cbranch_execz" for x in r):
            out.add(b)
    _cache[key] = out
    return out


def arrives_with_exec_zero(lines, lo, hi, i):
    """True when the block holding line i can only be entered with exec = 0 and nothing in the block
    before line i writes exec."""
    b = i
    while b > lo and not LABEL.match(lines[b]):
        if EXEC_WRITE.match(lines[b]):
            return False  # exec rewritten inside the block before line i
        b -= 1
    return b in zero_entry_blocks(lines, lo, hi)


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


VECTOR = re.compile(r"^\s*(v_|ds_|buffer_|global_|scratch_|flat_)")


def dead_vector_instructions(lines, lo, hi):
    """Every vector instruction that sits in a block entered with exec = 0 before anything rewrites exec:
    reloads (above), but also the register COPIES the allocator inserts (`v_mov_b32 vA, vB` when it splits a
    live range) -- under exec = 0 none of them does anything, so none of them can have been meant."""
    found = []
    for b in sorted(zero_entry_blocks(lines, lo, hi)):
        j = b + 1
        while j < hi and not LABEL.match(lines[j]) and not lines[j].startswith(".Lfunc_end"):
            t = lines[j]
            if EXEC_WRITE.match(t):
                break
            if VECTOR.match(t):
                found.append((j - lo, t.strip()))
            j += 1
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
    other = 0
    for name, lo, hi in kernels(lines):
        if key and key not in name:
            continue
        for off, ins in dead_vector_instructions(lines, lo, hi):
            if "Folded Reload" not in ins:
                other += 1
                print(f"{name} +{off}: {ins}   ** vector instruction in a block entered with exec = 0 **")
    print(f"{other} other vector instruction(s) in blocks entered with exec = 0")
    return 1 if (bad or other) else 0


if __name__ == "__main__":
    sys.exit(main())
