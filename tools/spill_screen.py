#!/usr/bin/env python3
"""Register-pressure screen of the frame kernel WITHOUT a GPU (profiles/r05_experiments.txt section 4).

Compiles ONE instantiation of render_frame_kernel to gfx950 assembly with the product flags (about 6 s with the
-DRM_ONLY_* switches of rm_kernels.hip) and prints what the allocator did: VGPRs, spilled VGPRs, scratch bytes per
lane, SGPRs, spilled SGPRs.  Each further argument is a set of extra compiler flags -- typically a -D that knocks a
source site out or swaps it for a stand-in -- screened against the plain build:

    python tools/spill_screen.py [--arith 3] [--layout 5] "" "-DRM_FRAME_MINW=6" "-DMY_KNOCKOUT=1" ...

A site that moves the spill count is a candidate for an EXACT rewrite, which is then timed on the GPU (tools/ab_build.py).
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raymarchcl_amd import _native  # noqa: E402


def screen(arith, layout, extra):
    flags = [f for f in _native.HIPCC_FLAGS if f not in ("-fPIC", "-shared")]
    flags += [f"-DRM_ONLY_ARITH={arith}", f"-DRM_ONLY_LAYOUT={layout}", "-DRM_ONLY_FRAME=1"] + extra.split()
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        r = subprocess.run([_native._hipcc()] + flags + ["--cuda-device-only", "-S", os.path.join(_native.CSRC, "rm_kernels.hip"),
                            "-o", out], capture_output=True, text=True)
        if r.returncode:
            return "does not compile: " + r.stderr.strip().splitlines()[-1][:160]
        text = open(out).read()
    for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)\.wavefront_size", text, re.S):
        if "render_frame_kernelILb1E" in m.group(1):
            v = {k: re.search(r"\." + k + r":\s+(\d+)", m.group(2)).group(1)
                 for k in ("vgpr_count", "vgpr_spill_count", "private_segment_fixed_size", "sgpr_count", "sgpr_spill_count")}
            return (f"vgpr {v['vgpr_count']} spilled {v['vgpr_spill_count']} scratch {v['private_segment_fixed_size']} B/lane  "
                    f"sgpr {v['sgpr_count']} spilled {v['sgpr_spill_count']}")
    return "frame kernel not found"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arith", type=int, default=3, help="contract (rm_math.hpp ArithOf): 0 cpu, 1 cpu + gpu casts, 2 strict, 3 default")
    ap.add_argument("--layout", type=int, default=5, help="table layout (rm_shade.hpp): 5 = 256^3, 3 = 512^3, 4 = 1024^3, 0-2 generic")
    args, variants = ap.parse_known_args()  # (every other argument -- "" or a quoted set of -D / -mllvm flags -- is a variant)
    for extra in variants or [""]:
        print(f"arith {args.arith} layout {args.layout} {extra or '(product)':40s} {screen(args.arith, args.layout, extra)}", flush=True)


if __name__ == "__main__":
    main()
