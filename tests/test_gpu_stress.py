"""Randomised parity against the CPU oracle inside the driver-run suite: bounded versions (seconds each) of the long stress
runs of tools/astar_stress.py, tools/astar_tiny_grids.py, tools/edt_stress.py and tools/toppra_stress.py (same case
generators, tests/stress_cases.py).  A* rounds assert the per-query expansion counts too, on both builds of the two-wavefront kernel and on
the one-wavefront kernel (SC_ASTAR_DUAL=0: what the overflow retry pass and sc_astar_gfield run)."""
import numpy as np
import pytest

import stress_cases as cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import sea_current_amd as sc
    c = sc.Context(0)
    yield c
    c.close()


def test_astar_random_maps_dual_kernel(ctx, oracle):
    rng = np.random.default_rng(2024)
    n = sum(cases.astar_round(ctx, oracle, rng, Q=96 if r % 8 else 3000, max_side=300, nthreads=16) for r in range(400))
    assert n >= 400 * 96 - 12 * 96


def test_astar_random_maps_single_wavefront_kernel(oracle, monkeypatch):
    import sea_current_amd as sc
    monkeypatch.setenv("SC_ASTAR_DUAL", "0")            # read when the context runs its first A*
    c = sc.Context(0)
    try:
        rng = np.random.default_rng(77)
        n = sum(cases.astar_round(c, oracle, rng, Q=96 if r % 8 else 3000, max_side=300, nthreads=16) for r in range(240))
        assert n >= 240 * 96 - 8 * 96
    finally:
        c.close()


def test_astar_random_maps_throughput_build_on_small_batches(oracle, monkeypatch):
    """Batches that fit the chip run the two-wavefront kernel's latency build (larger LDS ring), larger ones its throughput
    build; SC_ASTAR_LATENCY=0 puts the small random batches through the throughput build too."""
    import sea_current_amd as sc
    monkeypatch.setenv("SC_ASTAR_LATENCY", "0")         # read when the context runs its first A*
    c = sc.Context(0)
    try:
        rng = np.random.default_rng(515)
        n = sum(cases.astar_round(c, oracle, rng, Q=96, max_side=300, nthreads=16) for r in range(200))
        assert n >= 200 * 96 - 6 * 96
    finally:
        c.close()


def test_astar_tiny_grids(ctx, oracle):
    rng = np.random.default_rng(3)
    assert sum(cases.tiny_grid_round(ctx, oracle, rng) for _ in range(1000)) == 1000 * 40


def test_edt_random_shapes(ctx, oracle):
    rng = np.random.default_rng(7)
    assert sum(cases.edt_round(ctx, oracle, rng, max_cells=8_000_000) for _ in range(400)) > 100_000_000


def test_toppra_random_plans(ctx, oracle):
    rng = np.random.default_rng(11)
    tot = [cases.toppra_round(ctx, oracle, rng) for _ in range(400)]
    assert sum(t[1] for t in tot) > 1500        # feasible plans compared value by value
