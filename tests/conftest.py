"""pytest configuration.  `-m "not gpu"` runs on a CPU-only box; `-m gpu` needs one MI355X."""
import json
import os
import sys

import numpy as np
import pytest

# the library reads its tuning / test knobs (MRHIP_OPAIR, MRHIP_STREAM, ...) once per process unless told otherwise; tests
# switch kernels between calls with monkeypatch.setenv, so they ask for a fresh read every time
os.environ.setdefault("MRHIP_ENV_DYNAMIC", "1")
# the tuned FIRArbitrary kernels are tested at small sizes: keep small calls off the universal kernel here (api.hip: MRHIP_ARB_SMALL_MAX; the
# dispatch itself: test_small_arbitrary_calls_take_the_universal_kernel and the stress scripts, which run with the defaults)
os.environ.setdefault("MRHIP_ARB_SMALL_MAX", "0")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950) and the built HIP library")


@pytest.fixture(scope="session")
def pkg():
    """The product package (multirate.jl_amd/), loaded through __graft_entry__.load_package()."""
    import __graft_entry__ as ge
    lib = os.path.join(ge.PKG_DIR, "libmultirate_hip.so")
    if not os.path.exists(lib):
        ge.build()
    return ge.load_package()


@pytest.fixture(scope="session")
def O():
    """The CPU oracle (test infrastructure)."""
    from oracle import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def golden():
    d = os.path.join(ROOT, "tests", "golden")
    with open(os.path.join(d, "golden_v1.json")) as fh:
        meta = json.load(fh)
    return meta, np.load(os.path.join(d, "golden_v1.npz"))


@pytest.fixture(scope="session")
def known_answers():
    with open(os.path.join(ROOT, "tests", "golden", "reference_known_answers.json")) as fh:
        return json.load(fh)


def bits(a: np.ndarray) -> np.ndarray:
    """Reinterpret a float/complex array as unsigned integers for bit-exact comparison."""
    a = np.ascontiguousarray(a)
    if a.dtype in (np.float32, np.complex64):
        return a.view(np.uint32)
    return a.view(np.uint64)


def assert_bit_equal(a, b, what=""):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    assert a.dtype == b.dtype, f"{what}: dtype {a.dtype} vs {b.dtype}"
    if a.size:
        ba, bb = bits(a), bits(b)
        if not np.array_equal(ba, bb):
            bad = np.flatnonzero(ba != bb)
            raise AssertionError(f"{what}: {len(bad)} of {ba.size} words differ, first at {bad[0]}: "
                                 f"{a.reshape(-1)[bad[0] // (2 if np.iscomplexobj(a) else 1)]} vs "
                                 f"{b.reshape(-1)[bad[0] // (2 if np.iscomplexobj(b) else 1)]}")
