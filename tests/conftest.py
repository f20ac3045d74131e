import os as _os
# worker processes started by the tests inherit this: BLAS / OpenMP pools sized to 256 cores spin past the
# container's CPU quota and starve the kernel-launching threads (see bench.py)
for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "NUMEXPR_NUM_THREADS"):
    _os.environ.setdefault(_v, "1")
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_py
    return oracle_py.Oracle()


@pytest.fixture(scope="session")
def oracle_omp():
    """The oracle's OpenMP build (oracle/liboracle_omp.so) on 4 threads: the SAME statements as the bit-for-bit
    restatement with every sum split into per-thread partial sums.  Used only to measure the CONDITIONING FLOOR of a
    comparison: how far the reference algorithm itself moves when nothing but the association of its sums changes.
    A device (or sharded) run, whose reductions are necessarily grouped differently from the reference's sequential
    loops, cannot be held to less than a small multiple of that.  None when the library is missing."""
    import oracle_py
    try:
        o = oracle_py.Oracle(omp=True)
        o.set_threads(4)
        return o
    except OSError:
        return None


def floor_or(tol, floor, factor=3.0):
    """the tolerance BASELINE.json states (1e-10 relative), or `factor` x the measured conditioning floor if larger"""
    return max(tol, factor * (floor or 0.0))


@pytest.fixture(scope="session")
def reference():
    """The real reference templates (oracle/_ref/libref.so), when prebuilt."""
    import oracle_py
    if not oracle_py.have_reference():
        pytest.skip("oracle/_ref/libref.so not built (needs /root/reference)")
    return oracle_py.Reference()


@pytest.fixture(scope="session")
def golden():
    def load(name):
        with open(os.path.join(GOLDEN, name)) as f:
            return json.load(f)
    return load


@pytest.fixture(scope="session")
def ctx():
    """GPU context through the C ABI.  No fallback: fails when the library or the GPU is missing."""
    from optimization_amd import capi
    c = capi.Context(0)
    yield c
    c.close()


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = max(np.linalg.norm(b), 1e-300)
    return np.linalg.norm(a - b) / den


def trace_close(a, ref, floor_trace, tol, factor=3.0):
    """Per-iteration scalars (alpha_k, beta_k) against the reference's: |a_k / ref_k - 1| <= tol, or -- where the SAME
    reference algorithm with re-associated sums (floor_trace, conftest.oracle_omp) has itself moved further than that --
    `factor` x the running maximum of its deviation up to k.  Returns (ok, message with the measured pair)."""
    a, ref = np.asarray(a, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    k = min(a.size, ref.size)
    e = np.abs(a[:k] / ref[:k] - 1)
    bar = np.full(k, tol)
    fl = np.zeros(k)
    if floor_trace is not None:
        # (a list of traces = several re-associations, e.g. 2, 3 and 4 threads: their pointwise maximum -- one
        # realisation of the floor is itself a noisy estimate at the rounding end of a deep solve)
        many = isinstance(floor_trace, (list, tuple)) and len(floor_trace) and np.ndim(floor_trace[0]) > 0
        for ft in (floor_trace if many else [floor_trace]):
            fm = np.abs(np.asarray(ft, dtype=np.float64)[:k] / ref[:k] - 1)
            fl[:fm.size] = np.maximum(fl[:fm.size], np.maximum.accumulate(fm))
        bar = np.maximum(bar, factor * fl)
    bad = np.nonzero(e > bar)[0]
    msg = (f"max deviation {e.max():.2e} at k = {int(e.argmax())} (re-associated reference there: {fl[int(e.argmax())]:.2e}; "
           f"bar {tol:.0e} or {factor:g} x floor)" + (f"; first violation at k = {int(bad[0])}: {e[bad[0]]:.2e} > {bar[bad[0]]:.2e}"
                                                      if bad.size else ""))
    return bad.size == 0 and a.size == ref.size, msg
