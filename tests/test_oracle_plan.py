"""The oracle's restatement of PathTimingTrajectory::Plan (oracle/tp_oracle_plan.c) against the
reference's own planner tests (path_timing_trajectory_test.cc:112-296): waypoints (1,2,3),
(-1,-2,-3), (1,2,3); v_max 1, a_max 2; 1000 path samples (default sampling distance 0.005: the
path needs several windows); 4 ms time step; 750 ms horizon; replanning every 200 ms; both time
sampling methods. The reference's tests assert properties (no stored numbers exist for the
planner): they are asserted here on the oracle, which the HIP-backed mirror is then compared
with bit for bit (tests/cpp/test_host_api.cc)."""
import numpy as np
import pytest

from oracle import tpo

MS = 1_000_000
WAYPOINTS = np.array([[1.0, 2.0, 3.0], [-1.0, -2.0, -3.0], [1.0, 2.0, 3.0]])


def make(skip, N=1000):
    p = tpo.Planner(3, N, skip=skip)
    p.set_limits(np.full(3, 1.0), np.full(3, 2.0))
    p.set_waypoints(WAYPOINTS)
    return p


@pytest.mark.parametrize("skip", [False, True])
def test_rest_to_rest_planning_with_replans(skip):
    """RestToRestPlanningWorks (:112-173): Plan repeatedly, start shifted by the replan interval,
    until the trajectory is at its end; then the velocity is zero at the last waypoint."""
    p = make(skip)
    start, loops, total_windows = 0, 0, 0
    while not p.target_reached:
        assert p.plan(start, 750 * MS) == 0
        M = p.num_samples
        assert M > 0 and p.positions.shape == (M, 3) and p.velocities.shape == (M, 3)
        t = p.time
        assert abs(t[0] - start / 1e9) < 1e-12 and (np.diff(t) > 0).all()   # EXPECT_DOUBLE_EQ there
        assert (np.abs(p.velocities) <= 0.8 * 1.0 + 1e-9).all()       # constraint_safety 0.8
        assert (np.abs(p.accelerations) <= 2.0 + 1e-12).all()
        total_windows += p.windows
        start = min(p.end_time, start + 200 * MS)
        loops += 1
        assert loops < 200
    assert loops > 5 and total_windows >= 3                             # several windows were chained
    np.testing.assert_allclose(p.velocities[-1], 0.0, atol=1e-12)
    np.testing.assert_allclose(p.positions[-1], WAYPOINTS[-1], atol=1e-9)


@pytest.mark.parametrize("skip", [False, True])
def test_no_duplicate_initial_samples_and_start_alignment(skip):
    """NoDuplicateInitialSamples (:175-252), the parts that do not need TestOnlySetTimeSamples:
    the first sample sits exactly at the start time, the second one clearly after it, also
    when replanning from the time of an existing sample."""
    p = make(skip)
    eps = 0.01 * 0.004
    assert p.plan(0, 750 * MS) == 0
    t = p.time
    assert len(t) >= 5 and t[0] == 0.0 and t[1] >= t[0] + eps
    for sample in (0, 2, 1, 3):
        start = int(p.time[sample] * 1e9)              # TimeFromSec truncates
        assert p.plan(start, 750 * MS) == 0
        t = p.time
        assert len(t) >= 5 and abs(t[0] - start / 1e9) < 1e-12 and t[1] >= t[0] + eps


@pytest.mark.parametrize("skip", [False, True])
def test_is_invariant_to_starting_time(skip):
    """IsInvariantToStartingTime (:254-296), tolerance 1e-10 as there."""
    a, b = make(skip), make(skip)
    assert a.plan(0, 750 * MS) == 0 and b.plan(42_000 * MS, 750 * MS) == 0
    assert a.num_samples == b.num_samples
    np.testing.assert_allclose(a.positions, b.positions, atol=1e-10, rtol=0)


def test_argument_errors_follow_the_reference():
    """HandleTimeArguments (:502-538) and the missing-path case (:582-584)."""
    p = tpo.Planner(3, 1000)
    assert p.plan(0, 1000 * MS) == 1                    # kFailedPrecondition: no path
    p = make(False)
    assert p.plan(10_000 * MS, 750 * MS) == 0
    assert p.plan(9_000 * MS, 750 * MS) == 3            # start before the previous start
    far = p.end_time + 5 * MS
    assert p.plan(far, 750 * MS) == 2                   # beyond the previous plan + one time step


def test_already_planned_enough_erases_only(monkeypatch):
    """:596-601 with EraseTrajectoryBefore :540-575: with a long horizon already planned, a later
    Plan only drops the samples before the new start (both sampling methods)."""
    for skip in (False, True):
        p = make(skip)
        assert p.plan(0, 1_000_000 * MS) == 0 and p.target_reached     # whole path at once
        t0, q0 = p.time, p.positions
        assert p.plan(500 * MS, 100 * MS) == 0 and p.windows == 0
        t1 = p.time
        assert t1[0] == 0.5 and t1[-1] == t0[-1] and len(t1) < len(t0)
        if not skip:
            k = len(t0) - len(t1)
            np.testing.assert_array_equal(t1, t0[k:])
            np.testing.assert_array_equal(p.positions, q0[k:])
        else:
            # first sample interpolated at the start time, the rest are kept path samples
            np.testing.assert_array_equal(t1[1:], t0[len(t0) - len(t1) + 1:])
            assert t1[1] - t1[0] >= 0.95 * 0.004 - 1e-12
