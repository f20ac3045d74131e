"""Host-side logic of the product that needs no GPU: the run plan of the LDS-window kernels (sparse.hip window_plan /
window_runs through mi_debug_window_runs)."""
import numpy as np
import pytest

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
