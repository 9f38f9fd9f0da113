"""CPU: the vectorised host port behind bench.py's cpu_baseline (oracle/sfm_oracle_fast.c) gives exactly the counts, keys and
candidates of the scalar restatement (oracle/sfm_oracle.c) -- ordinary scenes, degenerate hypotheses, NaN / huge / generic-z
coordinates, thresholds inside and outside the division-free filter's range."""
import numpy as np
import pytest

import oracle as O
from cuda_sfm_amd_synth import synth


def scene(n, seed, **kw):
    sc = synth.two_view_scene(n, seed=seed, **kw)
    _, _, X0, X1 = O.fill_xu(sc["sift"], sc["Kinv"])
    return np.ascontiguousarray(X0), np.ascontiguousarray(X1)


@pytest.mark.parametrize("n,H,sweeps", [(8, 5, 0), (100, 64, 7), (1000, 300, 0), (4097, 40, 0), (2048, 257, 3)])
def test_fast_equals_scalar(n, H, sweeps):
    X0, X1 = scene(n, 5 + n)
    a = O.ransac_range(X0, X1, 3, H, 1e-6, sweeps, seed=9, want_E=True, nthreads=2)
    b = O.ransac_range_fast(X0, X1, 3, H, 1e-6, sweeps, seed=9, want_E=True, nthreads=2)
    assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2].view(np.uint32), b[2].view(np.uint32))


@pytest.mark.parametrize("thr", [1e-13, 1e-9, 1e-3, 10.0, 5e3])
def test_thresholds(thr):
    X0, X1 = scene(700, 3, noise_px=2.0)
    a = O.ransac_range(X0, X1, 0, 80, thr, 0, seed=1)
    b = O.ransac_range_fast(X0, X1, 0, 80, thr, 0, seed=1)
    assert a[0] == b[0] and np.array_equal(a[1], b[1])


def test_awkward_points_and_matrices():
    rng = np.random.default_rng(2)
    X0, X1 = scene(900, 8)
    X0 = X0.copy(); X1 = X1.copy()
    X1[0, 5] = np.nan; X0[1, 9] = np.inf; X1[:, 20:40] *= np.float32(1e9); X0[:, 50:90] *= np.float32(3.0)     # NaN, inf, huge, z != 1
    X1[:2, 100:140] = X1[:2, 100:101]                                                                           # many points on one spot
    Es = [np.zeros(9, np.float32), np.full(9, np.nan, np.float32), np.eye(3, dtype=np.float32).reshape(9),
          np.array([0, -1, 0, 1, 0, 0, 0, 0, 0], np.float32)] + [rng.normal(size=9).astype(np.float32) for _ in range(40)]
    with np.errstate(invalid="ignore", over="ignore"):
        for E in Es:
            for thr in (1e-6, 1e-2):
                assert O.count_inliers_fast(E, X0, X1, thr) == O.count_inliers(E.reshape(3, 3), X0, X1, thr, want_mask=False)[0]
