#!/usr/bin/env python3
"""Register-allocation lottery: compile semantically neutral variants of the hot kernel
(-DRM_K=<bit mask> in a scratch copy of csrc that carries the knobs) and rank them by where their
spills land (tools/isa_spills.py logic).  No GPU needed; the best few are then measured.

    python tools/ra_lottery.py <srcdir with knobs> <first mask> <last mask> [jobs]
"""
import os, re, subprocess, sys
from concurrent.futures import ThreadPoolExecutor

src, lo, hi = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
jobs = int(sys.argv[4]) if len(sys.argv) > 4 else 8
KEY = "render_frame_kernelILb1ELi7ELb0ELb0ELb0ELi0E"


def score(path):
    lines = open(path).read().split("\n")
    tot = {}
    for i, l in enumerate(lines):
        if re.match(r"^_Z\S*:", l) and (KEY in l or "lighting_wave" in l or "occlusion_wave" in l or "shadows_wave" in l):
            end = next(j for j in range(i, len(lines)) if lines[j].startswith(".Lfunc_end"))
            depth = 0
            callee = KEY not in l
            for m in lines[i:end]:
                mm = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?$", m)
                if mm:
                    d = re.search(r"Depth=(\d+)", mm.group(2) or "")
                    depth = int(d.group(1)) if d else 0
                    continue
                if re.match(r"^; %bb\.\d+:", m):
                    d = re.search(r"Depth=(\d+)", m)
                    depth = int(d.group(1)) if d else 0
                    continue
                if "scratch_load" in m or "scratch_store" in m:
                    dd = depth + (1 if callee else 0)
                    tot[dd] = tot.get(dd, 0) + 1
    return tot


def run(k):
    out = f"/tmp/vb/lot_{k}"
    os.makedirs(out, exist_ok=True)
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-fno-slp-vectorize", "-std=c++17",
                    "-ffp-contract=off", f"-DRM_K={k}", "--cuda-device-only", "-S", f"{src}/rm_kernels.hip", "-o",
                    f"{out}/k.s"], stderr=subprocess.DEVNULL)
    t = score(f"{out}/k.s")
    w = sum(n * (1.0 if d == 0 else 1.3 if d == 1 else 60.0) for d, n in t.items())
    return k, w, t


with ThreadPoolExecutor(jobs) as ex:
    res = list(ex.map(run, range(lo, hi + 1)))
for k, w, t in sorted(res, key=lambda r: r[1]):
    print(f"RM_K={k:4d}  score {w:8.1f}  by depth {dict(sorted(t.items()))}")
