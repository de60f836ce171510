#!/usr/bin/env python3
"""Where does the compiler put its spills?  Reads the gfx950 assembly of a kernel
(hipcc -S --cuda-device-only) and reports scratch loads / stores by loop depth, using the loop
annotations LLVM prints on basic-block labels ("in Loop: Header=.. Depth=N").  A spill at
depth 0 costs once per sample; one inside a march or walk loop every turn.

    python tools/isa_spills.py k.s <mangled-name-substring> [more substrings: callees...]
"""
import re
import sys


def analyse(lines, key):
    start = next((i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and key in l), None)
    if start is None:
        print(f"{key}: not found")
        return
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    depth = 0
    lanes = {}
    hist = {}
    n_ins = 0
    per_loop = {}
    cur = None
    for l in lines[start:end]:
        m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?$", l)
        if m:
            c = m.group(2) or ""
            d = re.search(r"Depth=(\d+)", c)
            h = re.search(r"Header=(BB\d+_\d+)", c)
            if "Loop Header" in c:
                depth = int(d.group(1)) if d else 0
                cur = m.group(1)[2:]
            elif d:
                depth = int(d.group(1))
                cur = h.group(1) if h else None
            else:
                depth, cur = 0, None
            continue
        if re.match(r"^; %bb\.\d+:", l):
            c = l
            d = re.search(r"Depth=(\d+)", c)
            h = re.search(r"Header=(BB\d+_\d+)", c)
            depth = int(d.group(1)) if d else 0
            cur = h.group(1) if h else None
            continue
        if not re.match(r"^\s+[a-z]", l):
            continue
        n_ins += 1
        if "v_readlane_b32" in l or "v_writelane_b32" in l:
            lanes.setdefault(depth, 0)
            lanes[depth] += 1
        if "scratch_load" in l or "scratch_store" in l:
            k = "load" if "scratch_load" in l else "store"
            hist.setdefault(depth, {"load": 0, "store": 0})[k] += 1
            if cur:
                per_loop.setdefault((depth, cur), 0)
                per_loop[(depth, cur)] += 1
    return n_ins, hist, lanes, per_loop


def report(lines, key):
    res = analyse(lines, key)
    if res is None:
        return
    n_ins, hist, lanes, per_loop = res
    print(f"{key}: {n_ins} instructions")
    for d in sorted(hist):
        print(f"  loop depth {d}: {hist[d]['load']} scratch loads, {hist[d]['store']} scratch stores")
    if lanes:
        print("  SGPR spill traffic (v_readlane / v_writelane) by depth: " +
              ", ".join(f"{d}: {lanes[d]}" for d in sorted(lanes)))
    deep = sorted(((d, h, n) for (d, h), n in per_loop.items() if d >= 3), reverse=True)
    if deep:
        print("  loops at depth >= 3 with spills: " + ", ".join(f"{h}@{d}:{n}" for d, h, n in deep[:24]))


def main():
    lines = open(sys.argv[1]).read().split("\n")
    for key in sys.argv[2:]:
        report(lines, key)


if __name__ == "__main__":
    main()
