"""pytest fixtures over oracle/pin.py -- the checkers of the two device contracts (live reference build for
gfx950, else its committed recordings under tests/golden/gfx950_<build>/; with neither a test FAILS)."""
import pytest

from oracle.pin import (BUILDS, CONTRACT_OF, METRIC_BUILD, RECORDED, SAMPLE_STRIDE, Checker, CheckerMissing,  # noqa: F401
                        FastReference, fixed_dir, input_digest, rel_err, sha, undefined_work_items)

FIXED = fixed_dir("strict")


@pytest.fixture(scope="session")
def pin(oracle_mod):
    """Checker of RM_CONTRACT_GFX950_STRICT."""
    return Checker(oracle_mod, "strict")


@pytest.fixture(scope="session")
def pin_default(oracle_mod):
    """Checker of RM_CONTRACT_GFX950_DEFAULT (the library default)."""
    return Checker(oracle_mod, "default")


@pytest.fixture(scope="session", params=list(BUILDS))
def pin_each(request, oracle_mod):
    """Both device contracts in turn: (checker, contract name for Context.set_contract)."""
    return Checker(oracle_mod, request.param), CONTRACT_OF[request.param]


@pytest.fixture(scope="session")
def fast_ref(oracle_mod):
    """The reference built with its own options: yardstick of BASELINE's 1e-4 metric (not a bit-exact checker)."""
    return FastReference(oracle_mod)
