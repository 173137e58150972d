"""The oracle's quaternion spline restatement (oracle/tp_oracle_quat.c) against what the
reference's own tests hold for it (splines/bsplineq_test.cc):
  * QuatExp against the Mathematica table (:99-171), IsApprox = 1e-12 relative;
  * Exp(Log(q)) = q and Log(Exp(q/|q|)) = q/|q| on the same inputs (:172-197);
  * a degree-1 quaternion spline interpolates its control points and equals piecewise slerp
    (LinearCaseWorks :309-344);
  * a degree-3 spline with control points a, a, b, b rotates monotonically about ONE axis from
    a to b (SlerpInterpolationForAABBCase :805-867);
  * out-of-range parameters are refused (EvalCurveParameterRange :284-307)."""
import json
import os

import numpy as np
import pytest

from oracle import tpo

EPS = 1e-12   # Eigen::NumTraits<double>::dummy_precision(), the tolerance of IsApprox


def approx(a, b, tol=EPS):
    a, b = np.asarray(a), np.asarray(b)
    return np.linalg.norm(a - b) <= tol * min(np.linalg.norm(a), np.linalg.norm(b)) + 1e-300


def qmul(a, b):
    w1, x1, y1, z1 = a
    w2, x2, y2, z2 = b
    return np.array([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                     w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2])


def slerp(t, a, b):
    d = float(np.dot(a, b))
    if d < 0:
        b, d = -b, -d
    th = np.arccos(min(d, 1.0))
    if th < 1e-12:
        return a
    return (np.sin((1 - t) * th) * a + np.sin(t * th) * b) / np.sin(th)


def test_quat_exp_matches_the_mathematica_table(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "quat_exp_golden.json")))
    assert len(g["input"]) == 15
    for q, e in zip(g["input"], g["exp"]):
        # the table prints 16 significant digits: IsApprox's 1e-12 as in the reference
        assert approx(tpo.quat_exp(q), e, 1e-12), (q, tpo.quat_exp(q), e)
        assert approx(tpo.quat_exp(tpo.quat_log(q)), q)
        n = np.asarray(q) / np.linalg.norm(q)
        assert approx(tpo.quat_log(tpo.quat_exp(n)), n)


def test_quat_power_is_the_fractional_rotation():
    axis = np.array([1.0, -2.0, 0.5]); axis /= np.linalg.norm(axis)
    for ang in (0.3, 1.7, 3.0):
        q = np.concatenate([[np.cos(ang / 2)], np.sin(ang / 2) * axis])
        for p in (0.0, 0.25, 1.0, 2.5):
            want = np.concatenate([[np.cos(p * ang / 2)], np.sin(p * ang / 2) * axis])
            assert np.allclose(tpo.quat_power(q, p), want, atol=1e-14)
    assert np.allclose(tpo.quat_power([1.0, 0, 0, 0], 0.37), [1, 0, 0, 0], atol=0)   # identity, |v| = 0 branch


def test_linear_quaternion_spline_is_piecewise_slerp():
    r = np.sqrt(0.5)
    pts = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1], [r, r, 0, 0], [r, 0, r, 0],
                    [r, 0, 0, r]], dtype=float)
    knots = tpo.make_uniform_knots(len(pts), 1)[1]
    for i in range(1, len(knots) - 1):
        q = tpo.bsplineq_eval_curve(knots, 1, pts, knots[i])
        want = pts[i - 1] if pts[i - 1][0] >= 0 else -pts[i - 1]
        assert approx(q, want) or approx(q, -want)
        if i < len(pts) - 2:
            for t in (0.1, 0.3, 0.5, 0.7, 0.9):
                u = knots[i] + t * (knots[i + 1] - knots[i])
                q = tpo.bsplineq_eval_curve(knots, 1, pts, u)
                s = slerp(t, pts[i - 1], pts[i])
                assert approx(q, s, 1e-10) or approx(q, -s, 1e-10)


def _angle_axis(a, b):
    conj = np.array([a[0], -a[1], -a[2], -a[3]])
    d = qmul(conj, b)
    if d[0] < 0:
        d = -d
    nv = np.linalg.norm(d[1:])
    ang = 2 * np.arctan2(nv, d[0])
    return ang, (d[1:] / nv if nv > 1e-14 else np.zeros(3))


@pytest.mark.parametrize("qa,qb", [
    ([.007, -.707, .707, .007], [.007, .707, -.707, .007]),
    ([.707, .707, 0, 0], [.707, 0, .707, 0]),
    ([.707, 0, .707, 0], [.707, .707, 0, 0]),
    ([.707, 0, .707, 0], [.707, 0, 0, .707]),
])
def test_aabb_cubic_spline_is_a_single_axis_rotation(qa, qb):
    qa, qb = np.array(qa, float), np.array(qb, float)
    pts = np.array([qa, qa, qb, qb])
    knots = tpo.make_uniform_knots(4, 3)[1]
    total, axis = _angle_axis(qa / np.linalg.norm(qa), qb / np.linalg.norm(qb))
    prev = 0.0
    for s in np.arange(0.1, 1.0001, 0.1):
        q = tpo.bsplineq_eval_curve(knots, 3, pts, min(s, 1.0))
        assert abs(np.linalg.norm(q) - 1) < 1e-12 and q[0] >= 0
        ang, ax = _angle_axis(qa / np.linalg.norm(qa), q)
        assert ang >= prev - 1e-12
        assert np.allclose(ax, axis, atol=1e-9) or ang < 1e-9
        prev = ang
    assert abs(prev - total) < 1e-6           # kEpsilon of the reference test


def test_parameter_range_and_pose_sampling():
    pts = np.tile([1.0, 0, 0, 0], (5, 1))
    knots = np.array([0, 0, 0, 1, 2, 3, 3, 3], float)
    tpo.bsplineq_eval_curve(knots, 2, pts, 0.0)
    tpo.bsplineq_eval_curve(knots, 2, pts, 3.0)
    for bad in (-0.1, 3.1):
        with pytest.raises(ValueError):
            tpo.bsplineq_eval_curve(knots, 2, pts, bad)
    # pose samples: translation spline + rotation spline; beyond knots.back() - delta the last
    # control pose is repeated (timeable_path_cartesian_spline.cc:488-503)
    rng = np.random.default_rng(1)
    rot = rng.normal(size=(5, 4)); rot /= np.linalg.norm(rot, axis=1, keepdims=True)
    tr = rng.normal(size=(5, 3))
    N, delta = 40, 0.1
    poses = tpo.sample_pose_spline(knots, tr, rot, 0.0, delta, N)
    for i in range(N):
        u = i * delta
        if u < 3.0 - delta:
            np.testing.assert_array_equal(poses[i, :3], tpo.eval_curve(knots, 2, tr, u)[1])
            np.testing.assert_array_equal(poses[i, 3:], tpo.bsplineq_eval_curve(knots, 2, rot, u))
            assert abs(np.linalg.norm(poses[i, 3:]) - 1) < 1e-12 and poses[i, 3] >= 0
        else:
            np.testing.assert_array_equal(poses[i, :3], tr[-1])
            np.testing.assert_array_equal(poses[i, 3:], rot[-1])
