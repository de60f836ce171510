#!/usr/bin/env python3
"""Loop tree of one kernel in hipcc's gfx950 assembly with instruction counts per loop body.

    python tools/isa_loops.py file.s 'render_frame_kernel<true, 7, false, 5, 2>' [--blocks]

Per loop (LLVM's "Loop Header: Depth=" comments): VALU / SALU / VMEM / LDS / SMEM instructions of the
blocks that belong to the loop itself (excl) and including its sub-loops (incl), scratch traffic.
"""
import re
import subprocess
import sys
from collections import defaultdict


def demangle(n):
    try:
        return subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
    except Exception:
        return n


def classify(ins):
    op = ins.split()[0]
    if op.startswith(("v_", "ds_")) and not op.startswith("ds_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("buffer_", "global_", "flat_")):
        return "vmem"
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, want = sys.argv[1], sys.argv[2]
    show_blocks = "--blocks" in sys.argv
    lines = open(path).read().split("\n")
    # kernel extents
    starts = [(i, m.group(1)) for i, l in enumerate(lines) for m in [re.match(r"^(_Z\S+):\s*(;.*)?$", l)] if m]
    ker = None
    for k, (i, name) in enumerate(starts):
        if want in demangle(name).replace("(anonymous namespace)::", ""):
            end = next((j for j in range(i, len(lines)) if lines[j].strip().startswith(".end_amdhsa_kernel")), len(lines))
            endf = next((j for j in range(i, len(lines)) if lines[j].strip().startswith("s_endpgm")), end)
            ker = (i, max(endf, i), name)
            break
    if not ker:
        sys.exit("kernel not found")
    lo, hi, name = ker
    # last s_endpgm of the function: search .Lfunc_end
    hi = next((j for j in range(lo, len(lines)) if lines[j].startswith(".Lfunc_end")), hi)
    print("#", demangle(name), f"lines {lo}-{hi}")
    block = "entry"
    blocks = []  # (label, header_or_None, depth, counts)
    cur = dict(label="entry", loop=None, depth=0, is_header=False, parents=[], c=defaultdict(int))
    blocks.append(cur)
    pending_comment = []
    for l in lines[lo + 1:hi]:
        m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?$", l)
        if m:
            cur = dict(label=m.group(1), loop=None, depth=0, is_header=False, parents=[], c=defaultdict(int))
            blocks.append(cur)
            rest = m.group(2) or ""
            pending_comment = [rest]
            mm = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", rest)
            if mm:
                cur["loop"], cur["depth"] = ".L" + mm.group(1), int(mm.group(2))
            mm = re.search(r"Parent Loop (BB\d+_\d+) Depth=(\d+)", rest)
            if mm:
                cur["parents"].append((".L" + mm.group(1), int(mm.group(2))))
            if "Loop Header: Depth=" in rest:
                mm = re.search(r"Loop Header: Depth=(\d+)", rest)
                cur["is_header"], cur["loop"], cur["depth"] = True, cur["label"], int(mm.group(1))
            continue
        s = l.strip()
        if s.startswith(";"):
            mm = re.search(r"Parent Loop (BB\d+_\d+) Depth=(\d+)", s)
            if mm:
                cur["parents"].append((".L" + mm.group(1), int(mm.group(2))))
            mm = re.search(r"Loop Header: Depth=(\d+)", s)
            if mm:
                cur["is_header"], cur["loop"], cur["depth"] = True, cur["label"], int(mm.group(1))
            mm = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", s)
            if mm and not cur["is_header"]:
                cur["loop"], cur["depth"] = ".L" + mm.group(1), int(mm.group(2))
            continue
        if not s or s.startswith(".") or s.startswith("//"):
            continue
        cur["c"][classify(s)] += 1
    # loop parent map from headers
    parent = {}
    depth = {}
    for b in blocks:
        if b["is_header"]:
            depth[b["label"]] = b["depth"]
            ps = sorted(b["parents"], key=lambda t: t[1])
            parent[b["label"]] = ps[-1][0] if ps else None
    excl = defaultdict(lambda: defaultdict(int))
    first_line = {}
    for b in blocks:
        key = b["loop"]
        for k, v in b["c"].items():
            excl[key][k] += v
    incl = {k: defaultdict(int, v) for k, v in excl.items()}
    for lp in sorted(depth, key=lambda x: -depth[x]):
        p = parent.get(lp)
        if p is not None or True:
            tgt = p
            for k, v in incl.get(lp, {}).items():
                incl.setdefault(tgt, defaultdict(int))[k] += v
    children = defaultdict(list)
    for lp in depth:
        children[parent.get(lp)].append(lp)
    order = {b["label"]: i for i, b in enumerate(blocks)}

    def show(lp, ind):
        e, n = excl.get(lp, {}), incl.get(lp, {})
        print(f"{'  ' * ind}{lp or 'kernel body':14s} d{depth.get(lp, 0)}  excl V{e.get('valu', 0):5d} S{e.get('salu', 0):5d} "
              f"M{e.get('vmem', 0):3d} L{e.get('lds', 0):3d} SM{e.get('smem', 0):3d} scr{e.get('scratch', 0):3d} W{e.get('wait', 0):3d}"
              f"   incl V{n.get('valu', 0):5d} S{n.get('salu', 0):5d} M{n.get('vmem', 0):3d} scr{n.get('scratch', 0):3d}")
        for c in sorted(children.get(lp, []), key=lambda x: order.get(x, 0)):
            show(c, ind + 1)

    show(None, 0)
    if show_blocks:
        for b in blocks:
            c = b["c"]
            print(f"{b['label']:14s} loop={b['loop']} d{b['depth']} V{c.get('valu',0)} S{c.get('salu',0)} M{c.get('vmem',0)} L{c.get('lds',0)} scr{c.get('scratch',0)}")


if __name__ == "__main__":
    main()
