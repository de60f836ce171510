#!/usr/bin/env python3
"""Command-line front of raymarchcl_amd/isa_exec_lint.py (the lint lives in the package: the build runs it).

    python tools/isa_exec_lint.py k.s [kernel-name-substring]      exit code 1 if anything is found
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raymarchcl_amd.isa_exec_lint import *  # noqa: F401,F403,E402
from raymarchcl_amd.isa_exec_lint import main  # noqa: E402

if __name__ == "__main__":
    sys.exit(main())
