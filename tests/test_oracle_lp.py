"""Oracle LP vs the reference's literal LPs, its brute-force cross-check and an
independent LP solver."""
import json
import os

import numpy as np
import pytest

from oracle import tpo


@pytest.fixture(scope="module")
def cases(golden_dir):
    return json.load(open(os.path.join(golden_dir, "lp_regression.json")))["cases"]


def _highs(a, b, lo, hi):
    """max sd2 s.t. lo <= a*sdd + b*sd2 <= hi, sd2 >= 0 with scipy's HiGHS."""
    from scipy.optimize import linprog
    a, b, lo, hi = map(np.asarray, (a, b, lo, hi))
    A_ub = np.concatenate([np.stack([b, a], 1), -np.stack([b, a], 1)])
    b_ub = np.concatenate([hi, -lo])
    res = linprog(c=[-1.0, 0.0], A_ub=A_ub, b_ub=b_ub, bounds=[(0, None), (None, None)],
                  method="highs")
    return res


def test_regression_lps_simplex_equals_bruteforce(cases):
    # time_optimal_path_timing_test.cc:1074-1087
    assert len(cases) == 5
    for c in cases:
        s = tpo.find_max_sd2_simplex(c["a"], c["b"], c["lower"], c["upper"])
        r = tpo.find_max_sd2_bruteforce(c["a"], c["b"], c["lower"], c["upper"])
        assert abs(s[0] - r[0]) <= 1e-8
        assert abs(s[2] - r[2]) <= 1e-8


def test_regression_lps_against_independent_solver(cases):
    for c in cases:
        s = tpo.find_max_sd2_simplex(c["a"], c["b"], c["lower"], c["upper"])
        res = _highs(c["a"], c["b"], c["lower"], c["upper"])
        assert res.status == 0
        assert abs(-res.fun - s[0]) <= 1e-7


def test_random_lps_simplex_equals_bruteforce():
    # time_optimal_path_timing_test.cc:703-736 (same distributions; numpy RNG, 4000 cases:
    # std::mt19937 + std::uniform_*_distribution streams are standard-library specific)
    rng = np.random.default_rng(12345)
    for _ in range(4000):
        n = int(rng.integers(2, 51))
        a = rng.uniform(-100, 100, n)
        b = rng.uniform(-100, 100, n)
        lo = rng.uniform(-10, 0, n)
        hi = rng.uniform(0, 10, n)
        s = tpo.find_max_sd2_simplex(a, b, lo, hi)
        r = tpo.find_max_sd2_bruteforce(a, b, lo, hi)
        assert abs(s[0] - r[0]) <= 1e-8
        assert abs(s[1] - r[1]) <= 1e-8
        assert abs(s[2] - r[2]) <= 1e-8


def test_random_lps_against_independent_solver():
    rng = np.random.default_rng(7)
    for _ in range(300):
        n = int(rng.integers(2, 31))
        a = rng.uniform(-100, 100, n)
        b = rng.uniform(-100, 100, n)
        lo = rng.uniform(-10, 0, n)
        hi = rng.uniform(0, 10, n)
        s = tpo.find_max_sd2_simplex(a, b, lo, hi)
        res = _highs(a, b, lo, hi)
        if res.status == 0:
            assert abs(-res.fun - s[0]) <= 1e-7
        else:  # unbounded -> saturates at kMaxSd2 (.cc:1218-1223)
            assert s[0] == tpo.KMAXSD2


def test_unbounded_and_empty_lps_return_kmaxsd2():
    # quirk Q4: all-zero rows (end padding of SamplePath) and unbounded problems
    z = np.zeros(14)
    s = tpo.find_max_sd2_simplex(z, z, -np.ones(14), np.ones(14))
    assert s == (tpo.KMAXSD2, 0.0, tpo.KMAXSD2)
    # only acceleration rows: sd2 unbounded along sdd = 0
    s = tpo.find_max_sd2_simplex([1.0, 2.0], [0.0, 0.0], [-1.0, -1.0], [1.0, 1.0])
    assert s == (tpo.KMAXSD2, 0.0, tpo.KMAXSD2)


def test_find_sdd_extremes():
    # .cc:638-695 on a box: -1 <= sdd <= 1, 0 <= sd2 <= 4 plus a coupling row
    a = [1.0, 0.0, 1.0]
    b = [0.0, 1.0, 1.0]
    lo = [-1.0, 0.0, -10.0]
    hi = [1.0, 4.0, 1.5]
    assert tpo.find_sdd_max(a, b, lo, hi, 0.0) == 1.0
    assert tpo.find_sdd_max(a, b, lo, hi, 1.0) == 0.5   # coupling row active
    assert tpo.find_sdd_min(a, b, lo, hi, 1.0) == -1.0
    assert tpo.find_sdd_max(a, b, lo, hi, 5.0) == 0.0   # infeasible sd2 -> 0
