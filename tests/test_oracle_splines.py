"""Oracle vs the reference's own spline fixtures."""
import json
import os

import numpy as np
import pytest

from oracle import tpo


@pytest.fixture(scope="module")
def golden(golden_dir):
    return json.load(open(os.path.join(golden_dir, "bspline_golden.json")))


def _sample(golden):
    knots = np.array(golden["knots"], float)
    pts = np.array(golden["control_points"], float)
    n = golden["num_samples"]
    du = (knots[-1] - knots[0]) / (n - 1)
    return knots, pts, n, du


def test_eval_curve_matches_mathematica_tables(golden):
    # splines/bspline_test.cc:756-771: error norm of EvalCurve against tables X, Y
    knots, pts, n, du = _sample(golden)
    vals = np.zeros((n, 2))
    for i in range(n):
        rc, v = tpo.eval_curve(knots, golden["degree"], pts, knots[0] + i * du)
        assert rc == 0
        vals[i] = v
    tol = golden["tolerance_error_norm"]
    assert np.linalg.norm(vals[:, 0] - golden["tables"]["X"]) <= tol
    assert np.linalg.norm(vals[:, 1] - golden["tables"]["Y"]) <= tol


def test_eval_curve_and_derivatives_match_mathematica_tables(golden):
    # splines/bspline_test.cc:773-838: squared error norms for value and 3 derivatives
    knots, pts, n, du = _sample(golden)
    vals = np.zeros((n, 4, 2))
    for i in range(n):
        rc, v = tpo.eval_curve_and_derivatives(knots, golden["degree"], pts, knots[0] + i * du, 4)
        assert rc == 0
        vals[i] = v
    tol = golden["tolerance_error_norm"]
    for k, (nx, ny) in enumerate([("X", "Y"), ("Xp", "Yp"), ("Xpp", "Ypp"), ("Xppp", "Yppp")]):
        assert np.sum((vals[:, k, 0] - golden["tables"][nx]) ** 2) <= tol, nx
        assert np.sum((vals[:, k, 1] - golden["tables"][ny]) ** 2) <= tol, ny
    # the tables are not trivially zero
    assert abs(golden["tables"]["Yppp"][-1]) == 168.0


def test_basis_partition_of_unity_degree2_repeated_knot():
    # splines/bspline_test.cc:1091-1129: degree 2, knots with a repeated interior knot
    knots = np.array([0, 0, 0, 1, 2, 3, 4, 4, 5, 5, 5], float)
    num_points = len(knots) - 3
    u = knots[0]
    while u <= knots[-1]:
        basis = np.zeros(num_points)
        for k in range(num_points):
            poly = np.zeros((num_points, 1))
            poly[k] = 1.0
            rc, v = tpo.eval_curve(knots, 2, poly, u)
            assert rc == 0
            basis[k] = v[0]
            assert basis[k] >= 0.0
        assert abs(basis.sum() - 1.0) <= 4 * np.finfo(float).eps
        u += 0.01


def test_uniform_knot_vectors_literal():
    # splines/bspline_test.cc:1270-1299
    rc, k = tpo.make_uniform_knots(8 - 3 - 1, 3)
    assert rc == 0 and list(k) == [0., 0., 0., 0., 1., 1., 1., 1.]
    rc, k = tpo.make_uniform_knots(9 - 3 - 1, 3)
    assert rc == 0 and list(k) == [0., 0., 0., 0., 0.5, 1., 1., 1., 1.]
    rc, k = tpo.make_uniform_knots(7 - 1 - 1, 1)
    assert rc == 0 and list(k) == [0., 0., 0.25, 0.5, 0.75, 1., 1.]


def test_knot_span_convention():
    # splines/bspline_base.cc:218-246: u in [knots[i], knots[i+1]); last knot -> last span
    knots = np.array([0, 0, 0, 1, 2, 3, 3, 3], float)
    assert tpo.knot_span(knots, 2, 0.0) == 2
    assert tpo.knot_span(knots, 2, 0.999) == 2
    assert tpo.knot_span(knots, 2, 1.0) == 3
    assert tpo.knot_span(knots, 2, 2.5) == 4
    assert tpo.knot_span(knots, 2, 3.0) == len(knots) - 2 - 2


def test_out_of_range_parameter_is_an_error():
    # splines/bspline.h:520-523, :550-553
    knots = np.array([0, 0, 0, 1, 1, 1], float)
    pts = np.array([[0.0], [1.0], [2.0]])
    assert tpo.eval_curve(knots, 2, pts, -0.1)[0] == 1
    assert tpo.eval_curve(knots, 2, pts, 1.1)[0] == 1
    assert tpo.eval_curve_and_derivatives(knots, 2, pts, 1.1, 3)[0] == 1
    assert tpo.eval_curve_and_derivatives(knots, 2, pts, 0.5, 4)[0] == 2  # der > degree


def test_derivatives_match_finite_differences():
    rng = np.random.default_rng(3)
    pts = rng.normal(size=(9, 3))
    _, knots = tpo.make_uniform_knots(9, 2)
    knots = knots * 2.5
    h = 1e-6
    for u in np.linspace(0.05, 2.45, 25):
        _, v = tpo.eval_curve_and_derivatives(knots, 2, pts, u, 3)
        _, vp = tpo.eval_curve(knots, 2, pts, u + h)
        _, vm = tpo.eval_curve(knots, 2, pts, u - h)
        np.testing.assert_allclose(v[1], (vp - vm) / (2 * h), rtol=1e-5, atol=1e-6)


def test_polyline_corner_rounding_structure():
    # splines/spline_utils.cc:47-102: 3W-2 points, corners kept, inner points on the segments
    corners = np.array([[0.0, 0.0], [1.0, 0.0], [1.0, 2.0]])
    out = tpo.polyline_to_bspline3_waypoints(corners, 0.2)
    assert out.shape == (7, 2)
    np.testing.assert_array_equal(out[0::3], corners)
    np.testing.assert_allclose(out[1], [0.2, 0.0])
    np.testing.assert_allclose(out[2], [0.8, 0.0])
    np.testing.assert_allclose(out[4], [1.0, 0.2])
    np.testing.assert_allclose(out[5], [1.0, 1.8])
    # spacing rule: segments shorter than 4*radius use a quarter of the segment
    short = tpo.polyline_to_bspline3_waypoints(np.array([[0.0], [0.4]]), 0.2)
    np.testing.assert_allclose(short[:, 0], [0.0, 0.1, 0.3, 0.4])
    one = tpo.polyline_to_bspline3_waypoints(np.array([[1.0, 2.0]]), 0.2)
    assert one.shape == (4, 2) and np.all(one == [1.0, 2.0])


def test_polyline_corner_rounding_reference_fixtures(golden_dir):
    """The literal corners and control points of the reference's corner-rounding tests
    (tests/golden/spline_utils_golden.json, mined from splines/spline_utils_test.cc:31-146). The
    oracle restates the vector variant (spline_utils.cc:47-102): on the cases without rotation the
    pose variant (:104-204) places its control points at the same translations."""
    import json
    import os
    fx = json.load(open(os.path.join(golden_dir, "spline_utils_golden.json")))
    used = 0
    for case in fx["cases"]:
        if any(c["angle"] != 0.0 for c in case["corners"]):
            continue
        corners = np.array([c["translation"] for c in case["corners"]])
        out = tpo.polyline_to_bspline3_waypoints(corners, case["translation_radius"])
        assert out.shape == (case["num_control_points"], 3)
        if len(corners) == 1:                 # OneCorner: the corner four times
            assert (out == corners[0]).all()
        else:
            np.testing.assert_array_equal(out[0::3], corners)
        for idx, pose in case["expected"].items():
            np.testing.assert_allclose(out[int(idx)], pose["translation"], atol=1e-9, err_msg=case["name"])
        if case["name"] == "ZeroRadius":      # the extra control points sit on the corners
            np.testing.assert_array_equal(out[1], corners[0])
            np.testing.assert_array_equal(out[2], corners[1])
        used += 1
    assert used == 4
