import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

REFERENCE_PRESENT = os.path.exists("/root/reference/resources/renderer.cl")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs /root/reference (build container only)")


def pytest_collection_modifyitems(config, items):
    skip_ref = pytest.mark.skip(reason="/root/reference is not present on this machine")
    for item in items:
        if "reference" in item.keywords and not REFERENCE_PRESENT:
            item.add_marker(skip_ref)


@pytest.fixture(scope="session")
def oracle_mod():
    """The CPU restatement (test infrastructure), built on demand."""
    import oracle

    oracle.build(ref=REFERENCE_PRESENT)
    return oracle


@pytest.fixture(scope="session")
def native():
    """The product library; built on demand (hipcc cross-compiles without a GPU)."""
    from raymarchcl_amd import _native

    _native.build()
    return _native


@pytest.fixture(scope="session")
def gpu_ctx(native):
    ctx = native.Context(0, contract="cpu")  # raises loudly when there is no gfx950 device; shared by the CPU-oracle checks
    yield ctx
    ctx.close()


@pytest.fixture
def cpu_contract(native):
    """Contexts created without an explicit contract render in the CPU-device contract for the
    duration of the test (the library's own default is RM_CONTRACT_GFX950): what the modules that
    check against the CPU oracle need."""
    old = native.DEFAULT_CONTRACT
    native.DEFAULT_CONTRACT = "cpu"
    yield
    native.DEFAULT_CONTRACT = old


def load_golden(name):
    import numpy as np

    z = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    return {k: z[k] for k in z.files}
