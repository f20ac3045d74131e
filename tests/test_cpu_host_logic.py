"""Host-side logic of the product that needs no GPU: the run plan of the LDS-window kernels (sparse.hip window_plan /
window_runs through mi_debug_window_runs)."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import ROOT
from optimization_amd import capi

TILE = 256  # rows per tile (4 slices of 64)


@pytest.mark.parametrize("far", [0, 300, 2560, 10_000, 15_876, 40_000, 1_000_003])
def test_window_run_plan_is_a_partition_within_the_budget(far):
    """Every (tiles, budget) pair gets a plan: strictly increasing run starts from 0 to ntiles, at most `budget`
    runs -- including the pairs no candidate of the cost model serves (budget < ntiles < 2 x the occupancy wish),
    which once left the planner without a plan."""
    for ntiles in list(range(1, 1400, 13)) + [3907, 7813, 31_250, 125_000]:
        for wgs in (1, 2, 3, 64, 255, 256, 340, 384, 385, 511, 512, 700, 768, 1000, 1024):
            b = capi.window_runs(ntiles, wgs, 256, far)
            assert b[0] == 0 and b[-1] == ntiles, (ntiles, wgs, far)
            assert np.all(np.diff(b) > 0), (ntiles, wgs, far)
            assert len(b) - 1 <= wgs, (ntiles, wgs, far, len(b) - 1)


def test_window_run_plan_of_the_bench_problems():
    """cfg2 (100^3: 3907 tiles, plane stride 10000 rows): 501 runs of 7 or 8 tiles whose starts follow multiples of
    D / 5 = 2000 rows to within half a tile; St(8e6,3) (200^3): 1000 runs of 31 or 32 tiles; without a far stride:
    equal runs."""
    b = capi.window_runs(3907, 1024, 256, 10_000)
    assert len(b) - 1 == 501 and set(np.diff(b)[:-1]) <= {7, 8} and np.diff(b)[-1] <= 8  # (the last run is the rest)
    assert np.abs(b[:-1] * TILE - np.arange(len(b) - 1) * 2000.0).max() <= TILE / 2
    b = capi.window_runs(31_250, 1024, 256, 40_000)
    assert len(b) - 1 == 1000 and set(np.diff(b)) <= {31, 32}
    assert np.abs(b[:-1] * TILE - np.arange(len(b) - 1) * 8000.0).max() <= TILE / 2
    b = capi.window_runs(3907, 1024, 256, 0)
    d = np.diff(b)
    assert len(set(d[:-1])) == 1 and d[-1] <= d[0] and 384 <= len(d) <= 1024
    # fewer tiles than the budget: one run per tile
    assert np.array_equal(capi.window_runs(60, 1024, 256, 0), np.arange(61))


def test_window_run_plan_prefers_more_workgroups_among_equal_costs():
    """plans within 2 % of the cheapest: the one with the most workgroups (latency hiding beyond the cache)"""
    few = len(capi.window_runs(31_250, 512, 256, 40_000)) - 1
    many = len(capi.window_runs(31_250, 1024, 256, 40_000)) - 1
    assert few <= 512 < many <= 1024


@pytest.mark.parametrize("n", [1, 2, 7, 24, 48, 72, 96])
def test_library_rayleigh_ritz_unit_has_the_bits_of_the_header_compiled_elsewhere(n):
    """mi_rayleigh_ritz (csrc/rr_host.cpp: the header-only solver of LinearAlgebra/DenseSymmetricEigen.h compiled by
    g++ -O3 -mavx2 -ffp-contract=off) against the same header inside tests/cpp/libharness_host.so (g++ -O2
    -march=x86-64-v3 -ffp-contract=off): identical Ritz values and vectors, bit for bit; and the identities of
    LOBPCG.h:53-62 (C'AC = Theta, C'BC = I)."""
    path = os.path.join(ROOT, "tests", "cpp", "libharness_host.so")
    if not os.path.exists(path):
        pytest.skip("harness not built")
    hz = C.CDLL(path)
    dp = C.POINTER(C.c_double)
    hz.hz_rayleigh_ritz.restype = C.c_int
    hz.hz_rayleigh_ritz.argtypes = [C.c_int, dp, dp, dp, dp]
    rng = np.random.default_rng(100 + n)
    M = rng.normal(size=(3 * n + 5, n))
    A = np.asfortranarray(M.T @ M * rng.uniform(.5, 2.0))
    K = rng.normal(size=(n, n))
    B = np.asfortranarray(np.diag(rng.uniform(.5, 50.0, n)) + .05 * (K @ K.T))
    L = capi.load()
    L.mi_rayleigh_ritz.argtypes = [C.c_int, dp, dp, dp, dp]
    th1, C1 = np.zeros(n), np.zeros((n, n), order="F")
    th2, C2 = np.zeros(n), np.zeros((n, n), order="F")
    assert L.mi_rayleigh_ritz(n, A.ctypes.data_as(dp), B.ctypes.data_as(dp), th1.ctypes.data_as(dp),
                              C1.ctypes.data_as(dp)) == 0
    assert hz.hz_rayleigh_ritz(n, A.ctypes.data_as(dp), B.ctypes.data_as(dp), th2.ctypes.data_as(dp),
                               C2.ctypes.data_as(dp)) == 0
    assert np.array_equal(th1, th2) and np.array_equal(C1, C2)
    assert np.all(np.diff(th1) >= 0)
    scale = np.abs(th1).max()
    assert np.abs(C1.T @ A @ C1 - np.diag(th1)).max() <= 1e-10 * scale
    assert np.abs(C1.T @ B @ C1 - np.eye(n)).max() <= 1e-10


def test_template_layer_is_clean_under_asan_and_ubsan():
    """SURVEY.md 5 (sanitizer / host-hardening target): the drop-in headers instantiated on a host vector -- TNT with and
    without preconditioner, GradientDescent, LSQR, TNLS with and without preconditioner, the generic LOBPCG path and its
    Rayleigh-Ritz -- built with g++ -fsanitize=address,undefined -fno-sanitize-recover=all and run; any report (heap
    error, leak, signed overflow, misaligned or out-of-bounds access ...) fails."""
    from optimization_amd import build as b
    try:
        exe = b.build_sanitize(run=True)   # raises with the sanitizer's report on any finding
    except b.SanitizerUnavailable as e:    # (no libasan / libubsan, or LeakSanitizer may not run here)
        import pytest
        pytest.skip(str(e))
    import os
    assert os.path.exists(exe)


def test_workload_generators_are_machine_independent():
    """the synthetic inputs come from mt19937_64 + own arithmetic (optimization_amd/wlgen.c), not from
    numpy's generators, libm's sin or LAPACK's QR -- so the SAME BYTES on every host.  Known answers: the published
    first and 10000th outputs of mt19937_64 seeded with 5489 (ISO C++ [rand.predef]: the 10000th consecutive invocation
    of a default-constructed std::mt19937_64 produces 9981545732273789042); SHA-256 of the generated arrays as made on
    the build machine (a host that generates other bytes fails here, not in a parity test)."""
    import hashlib
    from optimization_amd import workloads as wl
    raw = wl.mt19937_64_raw(5489, 10000)
    assert int(raw[0]) == 14514284786278117030 and int(raw[9999]) == 9981545732273789042
    u = wl.uniform_pm1(20260928, 3000)
    assert np.all(u >= -1.0) and np.all(u < 1.0)
    assert np.allclose(u[:3], [0.63845261, 0.88333058, -0.03266483], atol=1e-8)
    sha = lambda a: hashlib.sha256(np.ascontiguousarray(a, dtype=np.float64).tobytes()).hexdigest()  # noqa: E731
    assert sha(u) == "f562f724a203acaab36c01d4cc998338651843b643e57d4a465690b7b4f407f6"
    X = wl.random_stiefel(1000, 3)
    assert sha(X) == "49df7c05a398148b5d0838971d9c3f00efd8884d4f04bae229adfa13a5e9d72b"
    assert np.abs(X.T @ X - np.eye(3)).max() < 1e-14
    Xb, modes = wl.stiefel_bench_iterate(24, 20, 16, 3, eps=1e-2, seed=5)      # (smoke()'s iterate)
    assert sha(Xb) == "1bb470fab32a91499f5689bdb6329c142a53827d6bdb2ec1142333093faf4832"
    assert np.abs(Xb.T @ Xb - np.eye(3)).max() < 1e-14
    # the own sin against libm, over every argument the grids of the test-suite use
    L = wl._wl()[0]
    for den in (17, 21, 25, 101, 127, 201):
        for num in range(0, 4 * den + 1):
            assert abs(L.wl_sin_pi_ratio(num, den) - np.sin(np.pi * num / den)) < 2e-15   # (libm gets a rounded argument)
    # the eigenvector generator against the numpy one the other tests use
    v, _ = wl.laplacian_3d_eigvec(24, 20, 16, 1, 2, 1)
    Xe, _ = wl.stiefel_bench_iterate(24, 20, 16, 3, eps=0.0, seed=5)
    assert min(np.abs(Xe[:, j] - v).max() for j in range(3)) < 1e-14


def test_bench_dry_run_layers_plans_every_rank_with_the_real_planning_code():
    """bench.py --dry-run-layers N (r06): at N = 1 the line says what a --gpus N run would launch, probe, verify and time,
    in order, and every rank's halo plan comes from mi_csr_shard_plan (the host-side planning code of
    mi_csr_create_sharded) -- no GPU needed.  Slab partition of cfg4's 200^3 grid over 8 ranks: 25 planes and 1e6 rows per
    rank, one 200 x 200 plane of halo rows from each neighbour, symmetric."""
    import importlib.util
    import os
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    saved = os.dup(1)   # (bench.py points descriptor 1 at stderr when imported: put it back afterwards)
    try:
        b = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(b)
        d = b.dry_run_layers(8, "auto", 0)
    finally:
        import sys
        os.dup2(saved, 1)
        os.close(saved)
        sys.stdout = sys.__stdout__
    assert d["world"] == 8 and d["halo_plan_symmetric"]
    assert [l["layer"] for l in d["layers_in_order"]] == ["peer", "peer-separate", "peer-separate-rprime", "rccl", "rccl2"]
    for r, rec in enumerate(d["ranks"]):
        assert rec["rows"] == 1_000_000 and rec["z_planes"] == [25 * r, 25 * r + 25]
        assert rec["halo_rows_needed"]["from_rank_below"] == (40_000 if r > 0 else 0)
        assert rec["halo_rows_needed"]["from_rank_above"] == (40_000 if r < 7 else 0)
    d2 = b.dry_run_layers(2, "rccl", 0)
    assert [l["layer"] for l in d2["layers_in_order"]] == ["rccl", "rccl2"] and d2["ranks"][0]["rows"] == 1_000_000
