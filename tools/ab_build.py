#!/usr/bin/env python3
"""Build A/B variants of the library next to the product one (free, on the CPU box):

    python tools/ab_build.py name1="-DRM_X=0" name2="-DRM_Y=1 -DRM_Z=0" ...

writes raymarchcl_amd/libraymarch_hip_ab_<name>.so (git-ignored; travels with gpurun).
Bench one with  RAYMARCH_LIB=libraymarch_hip_ab_<name>.so python bench.py ...
"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raymarchcl_amd import _native  # noqa: E402

procs = []
for arg in sys.argv[1:]:
    name, _, flags = arg.partition("=")
    out = os.path.join(_native.HERE, f"libraymarch_hip_ab_{name}.so")
    cmd = [_native._hipcc()] + _native.HIPCC_FLAGS + flags.split() + \
          [os.path.join(_native.CSRC, s) for s in _native.SOURCES] + ["-o", out]
    procs.append((name, subprocess.Popen(cmd)))
for name, p in procs:
    print(name, "rc", p.wait())
