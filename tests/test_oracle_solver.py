"""Oracle solver vs the reference's scenario property tests
and planner behaviours (item 6)."""
import numpy as np
import pytest

import scenarios
from oracle import tpo

CASES = scenarios.all_cases()


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_scenario_properties(case):
    name, (A, B, lo, hi), s0, s1, sd0, meta = case
    n, c = A.shape
    p = tpo.Profile(n, c)
    assert p.setup(A, B, lo, hi, s0, s1, sd0, 0.0, 0.0) == tpo.OK
    assert p.optimize() == tpo.OK
    t, s, sd, sdd = p.time, p.s, p.sd, p.sdd
    # physical limits within kTiny (time_optimal_path_timing_test.cc:105-108, :192-197, ...)
    assert scenarios.max_violation(meta, s, sd, sdd) < tpo.KTINY
    # SolutionSatisfiesConstraints().ok() where the reference asserts it (:491, :552)
    if meta["kind"] in ("sine", "circle"):
        assert p.constraint_violations() == 0
    assert np.all(np.diff(t) >= 0)
    assert sd[-1] == 0.0 and s[0] == s0 and s[-1] == s1
    if meta["kind"] == "circle":
        ok, q_s, q_sd, _ = p.query(0.0)     # sd(t0) == sd_start exactly (:538-541)
        assert ok and q_sd == sd0
    if meta["kind"] == "curved":            # middle 30 % rides the velocity limit (:424-428)
        assert scenarios.curved_mid_segment_error(meta, s, sd) < tpo.KTINY


def test_scalar_straight_close_to_analytic_bang_bang():
    # 1-D, amax 1, vmax 0.5, s in [0,1]: accelerate 0.5 s, cruise 1.5 s, brake 0.5 s = 2.5 s
    A, B, lo, hi = scenarios.scalar_straight(2001, 0.5, 1.0)
    p = tpo.Profile(2001, 2)
    assert p.setup(A, B, lo, hi, 0.0, 1.0) == 0 and p.optimize() == 0
    assert abs(p.time[-1] - 2.5) < 2e-3
    assert abs(p.sd.max() - 0.5) < 1e-12


def test_setup_failure_order():
    A, B, lo, hi = scenarios.scalar_straight(10, 0.5, 1.0)
    p = tpo.Profile(10, 2)
    bad_hi = hi.copy(); bad_hi[3] = lo[3] - 1.0          # every row of sample 3 infeasible
    assert p.setup(A, B, lo, bad_hi, 0.0, 1.0) == 2       # .cc:174-182
    assert p.setup(A, B, lo, hi, 1.0, 1.0) == 3           # .cc:185
    assert p.setup(A, B, lo, hi, 0.0, 1.0, -0.1) == 4     # .cc:190
    one_bad = hi.copy(); one_bad[3, 0] = lo[3, 0]         # only one row lower == upper
    assert p.setup(A, B, lo, one_bad, 0.0, 1.0) == 5      # .cc:557 (quirk Q5)
    assert p.setup(A, B, lo, bad_hi, 1.0, 1.0) == 2       # bounds are checked first


def test_query_matches_samples_and_is_monotone():
    A, B, lo, hi = scenarios.circle(51, 0.0, np.pi, 2.0, 1.2, 1.0)
    p = tpo.Profile(51, 4)
    assert p.setup(A, B, lo, hi, 0.0, np.pi, 0.1) == 0 and p.optimize() == 0
    t, s, sd = p.time, p.s, p.sd
    for k in range(1, 50):
        ok, qs, qsd, _ = p.query(t[k])
        assert ok and abs(qs - s[k]) < 1e-12 and abs(qsd - sd[k]) < 1e-9
    ts = np.linspace(t[0], t[-1], 400)
    ss = [p.query(x)[1] for x in ts]
    assert np.all(np.diff(ss) >= -1e-15)
    assert p.query(t[-1] + 1.0)[1:] == (np.pi, 0.0, 0.0)
    assert p.previous_index(t[0] - 1.0) == -1 and p.previous_index(t[-1] + 1.0) == 50


def _plan_joint(waypoints, vmax, amax, N, t0=0.0):
    cps, knots = tpo.joint_fit_spline(np.asarray(waypoints, float), 0.2)
    delta = knots[-1] / (N - 1)
    r = tpo.time_joint_batch(knots[None], cps[None], np.array([vmax]), np.array([amax]),
                             0.0, delta, N, time_start=t0)
    assert r["status"][0] == 0
    return r, cps


def test_planner_reaches_last_waypoint_with_zero_velocity():
    # path_timing_trajectory_test.cc:112-173: 3-dof, waypoints (1,2,3),(-1,-2,-3),(1,2,3)
    wp = [[1, 2, 3], [-1, -2, -3], [1, 2, 3]]
    r, _ = _plan_joint(wp, [1.0] * 3, [2.0] * 3, 1000)
    np.testing.assert_allclose(r["q"][0, -1], wp[-1], atol=1e-12)
    assert np.all(r["qd"][0, -1] == 0.0) and r["sd"][0, -1] == 0.0
    assert np.all(np.abs(r["qd"][0]) <= 1.0 * 0.8 + 1e-9)
    assert np.all(np.abs(r["qdd"][0]) <= 2.0 + 1e-12)
    amax = np.array([2.0] * 3)
    ot, os_, osd, osdd, oq, oqd, oqdd = tpo.resample_uniform(
        r["t"][0], r["s"][0], r["sd"][0], r["sdd"][0], r["q"][0], r["qd"][0], r["qdd"][0],
        0.0, 0.004, amax)
    np.testing.assert_allclose(oq[-1], wp[-1], atol=1e-12)   # :167-172
    assert np.all(oqd[-1] == 0.0) and np.all(oqdd[-1] == 0.0)
    np.testing.assert_allclose(np.diff(ot), 0.004, rtol=0, atol=1e-12)
    # symmetric finite differences of position track the velocities (:412-437; the reference
    # accepts 1e-2 away from the path end and 1e-1 over the last 20 samples)
    fd = (oq[2:] - oq[:-2]) / (ot[2:] - ot[:-2])[:, None]
    err = np.abs(fd - oqd[1:-1])
    assert err[:-20].max() < 5e-2 and err[-20:].max() < 1e-1


def test_planner_is_invariant_to_start_time():
    # path_timing_trajectory_test.cc:254-296: 1e-10
    wp = [[1, 2, 3], [-1, -2, -3], [1, 2, 3]]
    r0, _ = _plan_joint(wp, [1.0] * 3, [2.0] * 3, 1000, t0=0.0)
    r1, _ = _plan_joint(wp, [1.0] * 3, [2.0] * 3, 1000, t0=123.456)
    np.testing.assert_allclose(r1["t"][0] - 123.456, r0["t"][0], atol=1e-10)
    for k in ("s", "sd", "sdd", "q", "qd", "qdd"):
        np.testing.assert_array_equal(r1[k], r0[k])


def test_end_padding_beyond_last_knot():
    # timeable_path_joint_spline.cc:300-313 + quirk Q4: samples past the spline end are the
    # last control point with zero derivatives; their LP saturates at kMaxSd2.
    cps, knots = tpo.joint_fit_spline(np.array([[0.0, 0.0], [1.0, 1.0], [2.0, 0.0]]), 0.2)
    N = 200
    delta = 1.5 * knots[-1] / (N - 1)
    q, q1, q2 = tpo.joint_sample_path(knots, cps, 0.0, delta, N)
    past = np.arange(N) * delta >= knots[-1] + delta
    assert past.any()
    np.testing.assert_array_equal(q[past], np.tile(cps[-1], (past.sum(), 1)))
    assert np.all(q1[past] == 0) and np.all(q2[past] == 0)
    A, B, lo, hi = tpo.joint_constraint_setup(q1, q2, [1.0, 1.0], [2.0, 2.0])
    i = int(np.argmax(past))
    assert tpo.find_max_sd2_simplex(A[i], B[i], lo[i], hi[i])[0] == tpo.KMAXSD2
    r = tpo.time_joint_batch(knots[None], cps[None], np.array([[1.0, 1.0]]),
                             np.array([[2.0, 2.0]]), 0.0, delta, N)
    assert r["status"][0] in (0, 7, 8, 9, 10)   # must terminate with a reference outcome


def test_oracle_reproduces_committed_regression_vectors(golden_dir):
    """tests/golden/solver_oracle_derived.npz (tools/make_solver_golden.py): oracle-generated, hence a
    regression pin of the oracle itself, not a parity pin against the reference."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        "make_solver_golden", os.path.join(os.path.dirname(golden_dir.rstrip("/")), "..", "tools",
                                           "make_solver_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    stored = dict(np.load(os.path.join(golden_dir, "solver_oracle_derived.npz")))
    sha = bytes(stored.pop("sha256")).decode()
    assert mod.digest(stored) == sha, "fixture file corrupted"
    now = mod.compute()
    assert sorted(now) == sorted(stored)
    for k in stored:
        np.testing.assert_array_equal(now[k], stored[k], err_msg=k)
