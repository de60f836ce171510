"""The C-ABI library builds for gfx950, loads on a CPU-only box, and exports
every symbol include/raymarch_hip.h declares.  No compute calls here."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "raymarch_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rm_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_exported(native):
    lib = ctypes.CDLL(native.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 15
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert sorted(native.EXPORTS) == names


def test_abi_version_and_error_string(native):
    L = native.lib()
    assert L.rm_abi_version() == 4
    assert isinstance(L.rm_last_error(), bytes)


def test_no_silent_cpu_fallback(native):
    """Without a device the product path must fail loudly, never route to the oracle."""
    if native.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(native.RmError) as ei:
        native.Context(0)
    assert "no CPU fallback" in str(ei.value)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "raymarchcl_amd")
    for dirpath, _dirs, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert not re.search(r"#\s*include\s*[<\"][^>\"]*(oracle/|cl_scalar\.h|rm_restate)", src), f
                assert "librm_restate" not in src and "libref_oracle" not in src, f


def test_source_digest_is_a_property_of_the_sources(native, tmp_path):
    """bench.source_digest() ties a PMC measurement (profiles/r06_pmc_traffic.json) to the build it was taken on.  It
    must come out the same on every box for the same tracked sources: independent of where the ROCm tree lives
    (the flags embed ROCM_PATH) and of git-ignored scratch files next to the kernel sources (round 5 lost
    roofline.traffic on the driver's box to exactly these two)."""
    import subprocess
    import sys

    import bench

    here = bench.source_digest()
    assert here == bench.source_digest("/opt/rocm-7.2.0") == bench.source_digest("/some/other/prefix")
    # another ROCM_PATH in the environment of a fresh process
    code = "import bench; print(bench.source_digest())"
    for rocm in ("/opt/rocm", "/nonexistent/rocm"):
        out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=dict(os.environ, ROCM_PATH=rocm),
                             capture_output=True, text=True, check=True).stdout.strip()
        assert out == here, (rocm, out, here)
    # a scratch file (csrc/_*: git-ignored) does not enter
    scratch = os.path.join(native.CSRC, "_digest_probe.hip")
    try:
        open(scratch, "w").write("// scratch\n")
        assert bench.source_digest() == here
        assert "_digest_probe.hip" not in native.kernel_source_files()
    finally:
        os.remove(scratch)
    # ... and the tracked sources do
    tracked = subprocess.run(["git", "ls-files", "raymarchcl_amd/csrc"], cwd=ROOT, capture_output=True, text=True)
    if tracked.returncode == 0 and tracked.stdout.strip():
        assert sorted(os.path.basename(f) for f in tracked.stdout.split()) == native.kernel_source_files()
